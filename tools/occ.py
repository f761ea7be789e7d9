import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch; torch.cuda.init()
import labenv; labenv.select("libsggan_lab.so")
import sggan_amd
L = ctypes.CDLL(sggan_amd.LIB_PATH)
out = (ctypes.c_int * 8)()
n = L.sgg_debug_occupancy(out, 8)
print("occupancy (blocks/CU): glds_fwd, glds_dgrad, reg_fwd, wgrad =", list(out)[:n])

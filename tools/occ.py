import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("SGG_LIB_PATH", os.path.join(ROOT, "sg-gan-tf2_amd", "libsggan_lab.so"))   # sgg_debug_* live in the lab build
import torch; torch.cuda.init()
import sggan_amd
L = ctypes.CDLL(sggan_amd.LIB_PATH)
out = (ctypes.c_int * 8)()
n = L.sgg_debug_occupancy(out, 8)
print("occupancy (blocks/CU): glds_fwd, glds_dgrad, reg_fwd, wgrad =", list(out)[:n])

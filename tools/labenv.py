"""tools/ only: pick the library build a measurement tool runs against.

The product package (sg-gan-tf2_amd/) reads no environment.  The A/B scripts in this directory select a variant build
(`python sg-gan-tf2_amd/build.py --variant X -D...`) or the ablation build (`--lab`, libsggan_lab.so) through the
SGG_LIB_PATH environment variable; a tool imports this module before its first kernel call and the choice is handed to the
package explicitly (`_abi.use_library`)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def select(default=None):
    """SGG_LIB_PATH if set, else `default` (a file name under sg-gan-tf2_amd/), else the in-tree libsggan.so."""
    from sggan_amd import _abi
    path = os.environ.get("SGG_LIB_PATH") or (os.path.join(ROOT, "sg-gan-tf2_amd", default) if default else None)
    if path:
        _abi.use_library(path)
    return path

"""GB/s of the colour -> class-index map (sgg_seg_class_map, segment_class.py:60-99) on a Cityscapes-size label image:
3 (RGB) or 4 (RGBA) bytes in, 1 byte out per pixel.   python tools/bench_seg.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import labenv; labenv.select()
import sggan_amd
from sggan_amd import kernels as K

for ch in (3, 4):
    for shape in ((1024, 2048), (8, 1024, 2048)):
        img = torch.randint(0, 256, shape + (ch,), dtype=torch.uint8, device="cuda")
        for _ in range(3):
            K.seg_class_map(img)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(50):
            K.seg_class_map(img)
        e.record(); torch.cuda.synchronize()
        us = s.elapsed_time(e) / 50 * 1e3
        px = img.numel() // ch
        print(f"{str(shape):18s} {ch} B/px in: {us:7.1f} us  {px * (ch + 1) / us / 1e3:7.0f} GB/s", flush=True)

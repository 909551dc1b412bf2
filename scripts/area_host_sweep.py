"""Randomised CPU check of the fused fractional-area walk (lp_area_core.h through lilliput_hip_area420_host) against the oracle: sizes up to
2048, thin and flat sources, 4:2:0 / 4:2:2 / 4:4:4, all eight orientations, Fit and Resize. No GPU. usage: area_host_sweep.py [seed] [images]
(round 3: seeds 1-4 x 40 images = 1 720 geometries that the kernels take, 0 mismatches)"""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from oracle import oracle as O
import lilliput_amd
from lilliput_amd import synth
from test_area_fused import _with_exif_orientation, _crop_plan, _jpeg
lib = C.CDLL(lilliput_amd.lib_path())
u8p = C.POINTER(C.c_uint8)
fn=lib.lilliput_hip_area420_host
fn.argtypes = [u8p,u8p,u8p,C.c_uint32,C.c_uint32]+[C.c_int]*10+[u8p]; fn.restype = C.c_int
rgb = synth.synth_rgb(3, 2048)
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv)>1 else 1)
bad=n=kept=0
for it in range(int(sys.argv[2]) if len(sys.argv)>2 else 60):
    w=int(rng.integers(5,2048)); h=int(rng.integers(2,2048))
    if rng.random()<0.3: w=int(rng.integers(5,40))
    if rng.random()<0.3: h=int(rng.integers(2,40))
    tw=int(rng.integers(1,min(w,300)+1)); th=int(rng.integers(1,min(h,300)+1))
    ss=int(rng.integers(3))
    data=_jpeg(rgb,w,h,ss,int(rng.integers(30,98)))
    planes=[np.ascontiguousarray(O.jpeg_decode_plane(data,c)) for c in range(3)]
    px=O.jpeg_decode(data)
    for o in range(1,9):
        for method in (O.FIT,O.RESIZE):
            exp=O.transform_static(px,o,tw,th,method,False)
            nw,nh,left,top,wpc,hpc=_crop_plan(O,w,h,o,tw,th,method)
            out=np.zeros((nh,nw,3),np.uint8)
            rc=fn(planes[0].ctypes.data_as(u8p),planes[1].ctypes.data_as(u8p),planes[2].ctypes.data_as(u8p),planes[0].shape[1],planes[1].shape[1],w,h,ss,o,left,top,wpc,hpc,nw,nh,out.ctypes.data_as(u8p))
            if rc==1: kept+=1; continue
            n+=1
            if exp.shape!=out.shape or not np.array_equal(exp,out):
                bad+=1; print("MISMATCH",w,h,tw,th,ss,o,method)
print("ran",n,"kept",kept,"bad",bad)

import sys, time; sys.path.insert(0,'/root/repo')
import numpy as np
from microimagelib_amd import api
from tests import fullsize_inputs as fi
img=fi.hash_volume(fi.FULL_SHAPE,1); psf=fi.psf_a()
api.decon_singleview(img,psf,1); outb=np.empty_like(img)
for it in (20,20):
    t=time.perf_counter(); out,rec=api.decon_singleview(img,psf,it,out=outb); dt=time.perf_counter()-t
    print(f"its {it}: whole {dt*1e3:.1f} ms; rec init {rec[6]*1e3:.1f} prep {rec[7]*1e3:.1f} loop {rec[8]*1e3:.1f} total {rec[9]*1e3:.1f} -> download+rest {1e3*(rec[9]-rec[6]-rec[7]-rec[8]):.1f}")
img2=fi.hash_volume(fi.FULL_SHAPE,3)
m=fi.registration_matrices()[1]
t=time.perf_counter(); o,tm,rec=api.reg3d(img,img2,i_tmx=m,reg_choice=0,input_tmx=True); dt=time.perf_counter()-t
print(f"reg3d choice 0 (2 uploads, resample, 1 download): {dt*1e3:.1f} ms")
t=time.perf_counter(); o,tm,rec=api.reg3d(img,img2,i_tmx=m,reg_choice=0,input_tmx=True); dt=time.perf_counter()-t
print(f"reg3d choice 0 again: {dt*1e3:.1f} ms")

import ctypes as C, os, sys
sys.path.insert(0, os.getcwd())
import torch
import raytracing_c_amd as rt
from raytracing_c_amd import ctypes_abi as abi
from raytracing_c_amd.configs import load_config
from tools.exp_small import LG_NAMES
names = LG_NAMES + ["LEAFMAJ_X","LEAFMAJ_L","LEAFMAJ_W","LEAFONE_X","LEAFCAM_X","LEAFHALF_X","LEAFHALF_L","LEAFHALF_W"]
assert rt.lib.rt_init(0) == 0
for cfg,(w,h,s,b) in (("helmet",(1920,1080,256,8)),("tower",(1920,1080,128,12))):
    hs,_ = load_config(cfg)
    d = rt.lib.rt_scene_upload(C.byref(hs.scene))
    accum = torch.zeros((h,w,3),dtype=torch.int64,device="cuda")
    p = abi.RT_Render_Params(w,h,s,b,0x1234ABCD,0,1,0,0)
    for i in range(2):
        accum.zero_(); assert rt.lib.rt_render_accumulate(d,C.byref(p),accum.data_ptr(),None)==0
    torch.cuda.synchronize()
    buf=(C.c_uint64*len(names))(); assert rt.lib.rt_get_ledger(buf,len(names))==0
    L=dict(zip(names,[int(v) for v in buf]))
    print(cfg, {k:L[k] for k in ("LEAF_X","LEAF_L","LEAF_CAM","LEAFCAM_X","LEAFMAJ_X","LEAFMAJ_L","LEAFMAJ_W","LEAFHALF_X","LEAFHALF_L","LEAFHALF_W","LEAFONE_X")})

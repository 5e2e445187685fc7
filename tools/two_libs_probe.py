import ctypes as C, os, sys, importlib
ROOT="/root/repo"; sys.path.insert(0, ROOT)
import numpy as np
def load(libpath, tag):
    os.environ["XLBHIP_LIB"]=libpath
    for m in [k for k in sys.modules if k.startswith("xlb_amd")]: del sys.modules[m]
    import xlb_amd
    from xlb_amd import ComputeBackend, PrecisionPolicy
    from xlb_amd.default_config import get_context
    from xlb_amd.grid import grid_factory
    from xlb_amd.operator.stepper import IncompressibleNavierStokesStepper
    pp=PrecisionPolicy.FP32FP32; vs=xlb_amd.velocity_set.D3Q19(pp, ComputeBackend.HIP); xlb_amd.init(vs, ComputeBackend.HIP, pp)
    ctx=get_context(); grid=grid_factory((512,512,512)); st=IncompressibleNavierStokesStepper(grid=grid, boundary_conditions=[])
    fields=st.prepare_fields()
    return tag, ctx, st, fields
order = sys.argv[1:]
objs=[load(f"{ROOT}/xlb_amd/lib/{n}.so", n) for n in order]
for rep in range(3):
    for tag,ctx,st,(f0,f1,bm,mm) in objs:
        st.run(f0,f1,bm,mm,1.0,4); ctx.sync()
        _,ms=st.run_timed(f0,f1,bm,mm,1.0,20)
        print(tag, round(ms/20,4), flush=True)

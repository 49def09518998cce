"""The register-poison kernel shared by tests/test_gpu_poison.py and tests/sweeps/range_model_check.py --poison: leaves a NaN pattern in
every architectural VGPR and every AGPR of the SIMDs it runs on (a 512-register asm kernel; 4096 workgroups cover the chip
several times).  A kernel that afterwards reads a register lane it never wrote produces NaN instead of plausible garbage."""
import ctypes
import os
import subprocess


def make_poison(workdir):
    import torch
    body = ["v_mov_b32 v255, 0x7ff80000"] + [f"v_mov_b32 v{k}, 0x7ff80000" for k in range(1, 255)] + [f"v_accvgpr_write_b32 a{k}, v255" for k in range(256)]
    clob = ", ".join(f'"v{k}"' for k in range(1, 256)) + ", " + ", ".join(f'"a{k}"' for k in range(256))
    src = ('#include <hip/hip_runtime.h>\nextern "C" __global__ void __launch_bounds__(256) poison(int* out) {\n  asm volatile("' + "\\n".join(body)
           + '" ::: ' + clob + ');\n  if (out && threadIdx.x == 999) out[0] = 1;\n}\n')
    with open(os.path.join(workdir, "p.hip"), "w") as fh:
        fh.write(src)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--genco", "--offload-arch=gfx950", "-O1", "-o", os.path.join(workdir, "p.co"), os.path.join(workdir, "p.hip")])
    hip = ctypes.CDLL("libamdhip64.so.7")
    mod, fn = ctypes.c_void_p(), ctypes.c_void_p()
    assert hip.hipModuleLoadData(ctypes.byref(mod), open(os.path.join(workdir, "p.co"), "rb").read()) == 0
    assert hip.hipModuleGetFunction(ctypes.byref(fn), mod, b"poison") == 0

    def run():
        nullp = ctypes.c_void_p(0)
        arr = (ctypes.c_void_p * 1)(ctypes.cast(ctypes.pointer(nullp), ctypes.c_void_p))
        st = torch.cuda.current_stream().cuda_stream
        assert hip.hipModuleLaunchKernel(fn, 4096, 1, 1, 256, 1, 1, 0, ctypes.c_void_p(st), arr, None) == 0
        torch.cuda.synchronize()
    return run

"""Hardware probe: does `buffer_load_dwordx4 ... lds` zero-fill LDS for out-of-range offsets?
Builds a tiny kernel with hipcc at run time (needs /opt/rocm on the GPU box)."""
import ctypes, os, subprocess, sys, tempfile
import torch
SRC = r'''
#include <hip/hip_runtime.h>
typedef __attribute__((address_space(3))) void lds_void;
extern "C" __global__ void probe(const unsigned int* x, unsigned int* y, int nbytes, int so) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  unsigned int* s = reinterpret_cast<unsigned int*>(smem);
  for (int i = threadIdx.x; i < 512; i += 64) s[i] = 0xdeadbeefu;
  __syncthreads();
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned int*>(x), 0, nbytes, 0x00020000);
  // lanes 0..31 in range, lanes 32..63 out of range (0x80000000)
  const int voff = threadIdx.x < 32 ? (int)(threadIdx.x * 16) : (int)0x80000000u;
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void*)(smem), 16, voff, so, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = threadIdx.x; i < 512; i += 64) y[i] = s[i];
}
extern "C" void launch(const unsigned int* x, unsigned int* y, int nbytes, int so, hipStream_t st) {
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 4096, st, x, y, nbytes, so);
}
'''
d = tempfile.mkdtemp()
open(os.path.join(d, "p.hip"), "w").write(SRC)
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", os.path.join(d, "p.hip"), "-o", os.path.join(d, "p.so")])
lib = ctypes.CDLL(os.path.join(d, "p.so"))
x = torch.arange(1, 1025, dtype=torch.int32, device="cuda")
y = torch.zeros(512, dtype=torch.int32, device="cuda")
lib.launch(ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(y.data_ptr()), 4096, 64, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
torch.cuda.synchronize()
v = y.cpu().tolist()
print("in-range lanes  (expect 17..): ", v[0:8])
print("out-of-range lanes (0 = zero-filled, -559038737 = untouched):", v[128:136], v[252:256])
print("beyond the instruction (untouched):", v[256:260])

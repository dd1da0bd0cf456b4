// antagonist.hip -- DIAGNOSTIC (not part of the product): kernels that each load ONE shared resource of a CU while the
// classification pipeline runs beside them on another stream, to show which resource the pipeline's kernels wait for
// when two sub-batches overlap ("which shared resource the two streams fight over ... is asserted,
// not shown").  rocprofv3 --pmc serialises dispatches, so counters cannot be taken with two kernels running; per-kernel
// durations from --kernel-trace with a known antagonist beside them can.
//
//   kind 0  SLEEP   s_sleep only: takes wave slots, nothing else
//   kind 1  SALU    dependent scalar integer ops (the scalar unit is one per CU)
//   kind 2  VALU32  dependent v_fma_f32 chains (vector issue)
//   kind 3  VALU64  dependent v_fma_f64 chains (the FP64 pipe)
//   kind 4  LDS     ds_read_b64 from a 2 KB block (LDS bandwidth / the LDS instruction queue)
//   kind 5  MEM     16-B loads striding through a 1-GB buffer (L2 misses, HBM, the vector memory path)
//   kind 6  REGS    SLEEP with 128 VGPRs held per wave (register-file capacity: a quarter of a SIMD's file per wave)
//
// Every wave loops until *stop != 0 or max_iters turns (an exit condition every wave reaches).  One wave per block.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

__global__ void __launch_bounds__(64) k_ant_sleep(const volatile int *stop, int max_iters)
{ for (int i = 0; i < max_iters; i++)
    { __builtin_amdgcn_s_sleep(32);
      if ((i & 15) == 0 && *stop) break;
    }
}

__global__ void __launch_bounds__(64) k_ant_salu(const volatile int *stop, int max_iters, int *sink)
{ int s = __builtin_amdgcn_readfirstlane((int)blockIdx.x);
  for (int i = 0; i < max_iters; i++)
    { for (int k = 0; k < 64; k++)
        { int t;
          asm volatile("s_mul_i32 %0, %1, 0x10dcd\n\ts_add_u32 %0, %0, 0x3039\n\ts_xor_b32 %0, %0, %1\n\ts_lshr_b32 %1, %0, 3\n\ts_add_u32 %1, %1, %0"
                       : "=&s"(t), "+s"(s));
        }
      if ((i & 3) == 0 && *stop) break;
    }
  if (s == 0x7fffffff) *sink = s;
}

__global__ void __launch_bounds__(64) k_ant_valu32(const volatile int *stop, int max_iters, float *sink)
{ float a = threadIdx.x*1e-3f, b = 1.0001f, c = 1e-7f, d = a+1.f;
  for (int i = 0; i < max_iters; i++)
    { for (int k = 0; k < 64; k++)
        { a = __builtin_fmaf(a,b,c); d = __builtin_fmaf(d,b,c); }
      if ((i & 3) == 0 && *stop) break;
    }
  if (a+d == 12345.f) *sink = a;
}

__global__ void __launch_bounds__(64) k_ant_valu64(const volatile int *stop, int max_iters, double *sink)
{ double a = threadIdx.x*1e-3, b = 1.0000001, c = 1e-9, d = a+1.;
  for (int i = 0; i < max_iters; i++)
    { for (int k = 0; k < 32; k++)
        { a = __builtin_fma(a,b,c); d = __builtin_fma(d,b,c); }
      if ((i & 3) == 0 && *stop) break;
    }
  if (a+d == 12345.) *sink = a;
}

__global__ void __launch_bounds__(64) k_ant_lds(const volatile int *stop, int max_iters, double *sink)
{ __shared__ double buf[256];
  for (int k = threadIdx.x; k < 256; k += 64) buf[k] = k;
  __syncthreads();
  double acc = 0.;
  int idx = threadIdx.x;
  for (int i = 0; i < max_iters; i++)
    {
#pragma unroll 4
      for (int k = 0; k < 32; k++)
        { acc += buf[idx]; idx = (idx+67) & 255; }
      if ((i & 3) == 0 && *stop) break;
    }
  if (acc == 12345.) *sink = acc;
}

__global__ void __launch_bounds__(64) k_ant_mem(const volatile int *stop, int max_iters, const uint4 *big, size_t n16, unsigned *sink)
{ size_t p = ((size_t)blockIdx.x*7919u*64u+threadIdx.x) % n16;
  unsigned acc = 0;
  for (int i = 0; i < max_iters; i++)
    { for (int k = 0; k < 8; k++)
        { const uint4 v = big[p];
          acc += v.x^v.w;
          p += 1048583u;                                    // a prime number of 16-B pieces: every load a new 16-MB-distant line
          if (p >= n16) p -= n16;
        }
      if ((i & 3) == 0 && *stop) break;
    }
  if (acc == 0x12345u) *sink = acc;
}

__global__ void __launch_bounds__(64) k_ant_regs(const volatile int *stop, int max_iters, float *sink)
{ float r[112];
#pragma unroll
  for (int k = 0; k < 112; k++) r[k] = threadIdx.x+k;
  for (int i = 0; i < max_iters; i++)
    { __builtin_amdgcn_s_sleep(32);
#pragma unroll
      for (int k = 0; k < 112; k++) asm volatile("" : "+v"(r[k]));        // keep all of them live across the loop
      if ((i & 15) == 0 && *stop) break;
    }
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < 112; k++) s += r[k];
  if (s == 12345.f) *sink = s;
}

// The stop flag lives in DEVICE memory and is set by a 4-byte copy on a private stream.  (First version: a flag in mapped host
// memory, polled by every wave -- with it the whole pipeline ran 50-200x slower beside 1024 SLEEPING waves: 75 M uncached reads
// per second across the host link starve every other memory request of the device.  CP_ANT_POLL=host keeps that variant.)
static int *g_stop_h = nullptr, *g_stop_d = nullptr, *g_stop_hd = nullptr;
static hipStream_t g_ctl = nullptr;
static int g_one = 1, g_zero = 0;
static void *g_sink = nullptr, *g_big = nullptr;
static const size_t BIG = (size_t)1 << 30;

extern "C" int ant_init(void)
{ if (g_stop_h) return 0;
  if (hipHostMalloc((void **)&g_stop_h,64,hipHostMallocMapped) != hipSuccess) return -1;
  *g_stop_h = 0;
  if (hipHostGetDevicePointer((void **)&g_stop_hd,g_stop_h,0) != hipSuccess) return -1;
  if (hipMalloc((void **)&g_stop_d,64) != hipSuccess || hipMemset(g_stop_d,0,64) != hipSuccess) return -1;
  if (hipStreamCreateWithFlags(&g_ctl,hipStreamNonBlocking) != hipSuccess) return -1;
  if (const char *e = getenv("CP_ANT_POLL")) if (!strcmp(e,"host")) g_stop_d = g_stop_hd;
  if (hipMalloc(&g_sink,4096) != hipSuccess || hipMalloc(&g_big,BIG) != hipSuccess) return -1;
  (void)hipMemset(g_big,1,BIG);
  return hipDeviceSynchronize() == hipSuccess ? 0 : -1;
}

// launches `nwaves` single-wave blocks of the given kind on `stream`; they run until ant_stop() or max_iters turns
extern "C" int ant_launch(int kind, int nwaves, int max_iters, void *stream)
{ hipStream_t st = (hipStream_t)stream;
  __atomic_store_n(g_stop_h,0,__ATOMIC_RELEASE);
  if (g_stop_d != g_stop_hd)
    { if (hipMemcpyAsync(g_stop_d,&g_zero,4,hipMemcpyHostToDevice,g_ctl) != hipSuccess || hipStreamSynchronize(g_ctl) != hipSuccess) return -1; }
  switch (kind)
    { case 0: hipLaunchKernelGGL(k_ant_sleep,dim3(nwaves),dim3(64),0,st,g_stop_d,max_iters); break;
      case 1: hipLaunchKernelGGL(k_ant_salu,dim3(nwaves),dim3(64),0,st,g_stop_d,max_iters,(int *)g_sink); break;
      case 2: hipLaunchKernelGGL(k_ant_valu32,dim3(nwaves),dim3(64),0,st,g_stop_d,max_iters,(float *)g_sink); break;
      case 3: hipLaunchKernelGGL(k_ant_valu64,dim3(nwaves),dim3(64),0,st,g_stop_d,max_iters,(double *)g_sink); break;
      case 4: hipLaunchKernelGGL(k_ant_lds,dim3(nwaves),dim3(64),0,st,g_stop_d,max_iters,(double *)g_sink); break;
      case 5: hipLaunchKernelGGL(k_ant_mem,dim3(nwaves),dim3(64),0,st,g_stop_d,max_iters,(const uint4 *)g_big,BIG/16,(unsigned *)g_sink); break;
      case 6: hipLaunchKernelGGL(k_ant_regs,dim3(nwaves),dim3(64),0,st,g_stop_d,max_iters,(float *)g_sink); break;
      default: return -1;
    }
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

extern "C" void ant_stop(void)
{ if (!g_stop_h) return;
  __atomic_store_n(g_stop_h,1,__ATOMIC_RELEASE);
  if (g_stop_d != g_stop_hd) { (void)hipMemcpyAsync(g_stop_d,&g_one,4,hipMemcpyHostToDevice,g_ctl); (void)hipStreamSynchronize(g_ctl); }
}

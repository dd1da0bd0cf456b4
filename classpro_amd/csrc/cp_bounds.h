// cp_bounds.h -- -DCP_BOUNDS diagnostic build (scripts/bounds_check.sh): every access of the per-read kernels to a read's
// counts and bases goes through a checked view that knows the read's length; an index outside the read is COUNTED (and
// answered with 0), never dereferenced, and cp_debug_bounds() reports the count and the first offender.  The product
// build compiles the same code with plain pointers: CP_PROF_T = const uint16_t *, CP_SEQ_T = const char *.
// Why: a read's result must be a function of the read alone (hazard 8); the sanitizer job covers the scalar
// code the host can compile, this build the device-only code (LDS-window fills, wide loads, the seed kernel's loads).
#pragma once
#include <stdint.h>

#ifdef CP_BOUNDS
#ifdef __HIPCC__
extern __device__ unsigned long long g_bounds[4];          // hits, first offender: kind, index, length
#endif
CP_HDM void cp_bounds_hit(int kind, long long i, long long n)
{
#ifdef __HIP_DEVICE_COMPILE__
  if (atomicAdd(&g_bounds[0],1ull) == 0ull) { g_bounds[1] = (unsigned long long)kind; g_bounds[2] = (unsigned long long)i; g_bounds[3] = (unsigned long long)n; }
#else
  (void)kind; (void)i; (void)n;
#endif
}
template <class T, int KIND>
struct cp_chk_view
  { const T *p; int n;
    CP_HDM cp_chk_view() : p(nullptr), n(0) {}
    CP_HDM cp_chk_view(const T *q, int len) : p(q), n(len) {}
    CP_HDM T operator[](int i) const
    { if ((unsigned)i >= (unsigned)n) { cp_bounds_hit(KIND,i,n); return (T)0; }
      return p[i];
    }
    // a run of `len` elements from lo, for the wide loads: the pointer, or a counted miss (and a pointer to the start)
    CP_HDM const T *span(int lo, int len) const
    { if (lo < 0 || lo+len > n) { cp_bounds_hit(KIND+8,lo,n); return p; }
      return p+lo;
    }
  };
#define CP_BCHK(kind,i,n) do { if ((unsigned long long)(long long)(i) >= (unsigned long long)(long long)(n)) cp_bounds_hit((kind),(i),(n)); } while (0)
typedef cp_chk_view<uint16_t,1> CP_PROF_T;
typedef cp_chk_view<char,2>     CP_SEQ_T;
#define CP_PROF_VIEW(ptr,len) CP_PROF_T((ptr),(len))
#define CP_SEQ_VIEW(ptr,len)  CP_SEQ_T((ptr),(len))
template <class T, int KIND> CP_HDM const T *cp_span(const cp_chk_view<T,KIND> &v, int lo, int len) { return v.span(lo,len); }
template <class T> CP_HDM const T *cp_span(const T *v, int lo, int len) { (void)len; return v+lo; }      // unchecked callers (whole-batch kernels)
#define CP_SPAN(v,lo,len)     cp_span((v),(lo),(len))
// eight bases from `pos` in direction `dir` as one word (cp_ctx.h): checked like every other wide load
CP_HDM int cp_seq_dirword(const cp_chk_view<char,2> &seq, int rlen, int pos, int dir, uint64_t *D)
{ const int lo = dir > 0 ? pos : pos-7;
  if (lo < 0 || lo+8 > rlen) return 0;
  uint64_t x;
  __builtin_memcpy(&x,seq.span(lo,8),8);
  *D = dir > 0 ? x : __builtin_bswap64(x);
  return 8;
}
#else
#define CP_BCHK(kind,i,n) ((void)0)
typedef const uint16_t *CP_PROF_T;
typedef const char     *CP_SEQ_T;
#define CP_PROF_VIEW(ptr,len) (ptr)
#define CP_SEQ_VIEW(ptr,len)  (ptr)
#define CP_SPAN(v,lo,len)     ((v)+(lo))
#endif

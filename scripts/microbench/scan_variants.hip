// scan_variants.hip -- DIAGNOSTIC: where k_scan_candidates (classpro_amd/csrc/kernels.hip) loses against a pure streaming read.
// A read-only 16-B-per-lane non-temporal kernel reaches 7.0 TB/s on 4 GB here (scripts/microbench/read_bw.hip); the scan
// kernel 5.2.  This file times the product's kernel form (V0) next to forms that leave parts of it out or do them
// differently, on 2e9 random counts (4.0 GB: one bench launch), and checks every complete form's bitmap against V0's.
//   hipcc --offload-arch=gfx950 -O3 scripts/microbench/scan_variants.hip -o build_diag/scan_variants
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>
#include <cstring>

#define WAVE 64
typedef unsigned cp_u4v __attribute__((ext_vector_type(4)));
typedef unsigned short cp_us2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }
__device__ __forceinline__ int wave_from_below(int v, int first) { return __builtin_amdgcn_update_dpp(first,v,0x138,0xf,0xf,false); }
__device__ __forceinline__ int wave_of_last(int v) { return __builtin_amdgcn_readlane(v,WAVE-1); }
__device__ __forceinline__ uint4 nt_load(const uint4 *p)
{ cp_u4v x = __builtin_nontemporal_load(reinterpret_cast<const cp_u4v *>(p)); return make_uint4(x.x,x.y,x.z,x.w); }

__device__ __forceinline__ unsigned scan_pair_flags(unsigned prevcur, unsigned cur, int rep)
{ const cp_us2 a = __builtin_bit_cast(cp_us2,prevcur), b = __builtin_bit_cast(cp_us2,cur);
  const cp_us2 mn = __builtin_elementwise_min(a,b), mx = __builtin_elementwise_max(a,b);
  const cp_us2 r2 = { (unsigned short)rep, (unsigned short)rep }, two = { 2, 2 }, one = { 1, 1 };
  const cp_us2 s1 = __builtin_elementwise_sub_sat(r2,mn);
  const cp_us2 s2 = __builtin_elementwise_sub_sat((cp_us2)(mx-mn),two);
  const cp_us2 m = __builtin_elementwise_min(__builtin_elementwise_min(s1,s2),one);
  return __builtin_bit_cast(unsigned,m);
}
__device__ __forceinline__ unsigned group_bits(const uint4 v, unsigned before, int rep)
{ const unsigned w0 = scan_pair_flags(__builtin_amdgcn_alignbit(v.x,before,16),v.x,rep);
  const unsigned w1 = scan_pair_flags(__builtin_amdgcn_alignbit(v.y,v.x,16),v.y,rep);
  const unsigned w2 = scan_pair_flags(__builtin_amdgcn_alignbit(v.z,v.y,16),v.z,rep);
  const unsigned w3 = scan_pair_flags(__builtin_amdgcn_alignbit(v.w,v.z,16),v.w,rep);
  const unsigned t = w0 | (w1 << 2) | (w2 << 4) | (w3 << 6);
  return (t & 0x55u) | ((t >> 15) & 0xaau);
}

// MODE 0: the product's kernel.  1: no stores (kept alive by an impossible condition).  2: loads only (xor).
// 3: dword stores -- lane 4q gathers the bytes of lanes 4q..4q+3 by DPP and stores them as one dword.
// 4: as 0 with the predecessor of a wave's first row taken from the 16-byte load of the group before it instead of a
//    dependent 2-byte load.
template <int MODE, int UNROLL>
__global__ void __launch_bounds__(256)
k_scan(const uint16_t *__restrict__ prof, int64_t total, int rep, uint8_t *__restrict__ bitmap, int64_t nbytes)
{ const int64_t ngroups = total >> 3;
  const int64_t nthreads = (int64_t)gridDim.x*blockDim.x;
  const int lane = lane_id();
  const uint4 *vp = reinterpret_cast<const uint4 *>(prof);
  const int64_t wavesz = (int64_t)WAVE*UNROLL;
  const int64_t nwaves = nthreads/WAVE;
  const int64_t wid = ((int64_t)blockIdx.x*blockDim.x+threadIdx.x)/WAVE;
  unsigned acc = 0;
  for (int64_t base = wid*wavesz; base < ngroups; base += nwaves*wavesz)
    { uint4 v[UNROLL];
      int64_t g[UNROLL];
#pragma unroll
      for (int u = 0; u < UNROLL; u++)
        { g[u] = base+(int64_t)u*WAVE+lane;
          v[u] = (g[u] < ngroups) ? nt_load(&vp[g[u]]) : make_uint4(0,0,0,0);
        }
      if (MODE == 2)
        {
#pragma unroll
          for (int u = 0; u < UNROLL; u++) acc ^= v[u].x^v[u].y^v[u].z^v[u].w;
          continue;
        }
      unsigned carry = 0;
      if (lane == 0)
        carry = (base > 0) ? prof[base*8-1] : (v[0].x & 0xffff);
#pragma unroll
      for (int u = 0; u < UNROLL; u++)
        { const bool live = g[u] < ngroups;
          const unsigned before = (unsigned)wave_from_below((int)v[u].w,(int)(carry << 16));
          carry = (unsigned)wave_of_last((int)v[u].w) >> 16;
          const unsigned bits = group_bits(v[u],before,rep);
          if (MODE == 1) { acc += bits; }
          else if (MODE == 3)
            { // bytes of lanes 4q+1, 4q+2, 4q+3 into lane 4q: quad_perm DPP moves (row-local, no LDS)
              const unsigned b1 = (unsigned)__builtin_amdgcn_mov_dpp((int)bits,0x39,0xf,0xf,true);   // quad_perm [1,2,3,0]
              const unsigned b2 = (unsigned)__builtin_amdgcn_mov_dpp((int)bits,0x4e,0xf,0xf,true);   // quad_perm [2,3,0,1]
              const unsigned b3 = (unsigned)__builtin_amdgcn_mov_dpp((int)bits,0x93,0xf,0xf,true);   // quad_perm [3,0,1,2]
              const unsigned dw = bits | (b1 << 8) | (b2 << 16) | (b3 << 24);
              if ((lane & 3) == 0)
                { if (g[u]+3 < ngroups) __builtin_nontemporal_store(dw,reinterpret_cast<unsigned *>(bitmap+g[u]));
                  else for (int k = 0; k < 4; k++) if (g[u]+k < ngroups) bitmap[g[u]+k] = (uint8_t)(dw >> (8*k));
                }
            }
          else if (MODE == 5) { if (live) bitmap[g[u]] = (uint8_t)bits; }
          else if (MODE == 6) { if (u == 0) { if (live) __builtin_nontemporal_store((uint8_t)bits,&bitmap[g[u]]); } else acc += bits; }   // a quarter of the stores
          else if (MODE == 8) { if ((((base >> 8)+u) & ((rep >> 16)-1)) == 0) { if (live) __builtin_nontemporal_store((uint8_t)bits,&bitmap[g[u]]); } else acc += bits; }  // one row in (rep>>16)
          else if (MODE == 7) { if (bits) __builtin_nontemporal_store((uint8_t)bits,&bitmap[g[u]]); }                           // only the non-zero bytes
          else if (live)
            __builtin_nontemporal_store((uint8_t)bits,&bitmap[g[u]]);
        }
    }
  if ((MODE == 1 || MODE == 2 || MODE == 6 || MODE == 8) && acc == 0x12345678u) bitmap[0] = (uint8_t)acc;
  if (blockIdx.x == 0 && threadIdx.x == 0 && MODE != 1 && MODE != 2)
    { int64_t p0 = ngroups << 3;
      if (p0 < total)
        { unsigned bits = 0;
          for (int64_t p = p0; p < total; p++)
            if (p > 0)
              { unsigned a = prof[p-1], b = prof[p];
                unsigned mn = a < b ? a : b, df = a < b ? b-a : a-b;
                if (mn < (unsigned)rep && df >= 3) bits |= 1u << (p-p0);
              }
          bitmap[ngroups] = (uint8_t)bits;
          p0 += 8;
        }
      for (int64_t q = p0 >> 3; q < nbytes; q++) bitmap[q] = 0;
    }
}

// V5: a lane takes TWO consecutive groups (32 B: two 16-byte loads 16 B apart, a wave row = 2 KB) and stores 2 bytes;
// the predecessor of the second group is the lane's own first group.  Half the cross-lane moves and stores per byte read.
template <int UNROLL>
__global__ void __launch_bounds__(256)
k_scan2(const uint16_t *__restrict__ prof, int64_t total, int rep, uint8_t *__restrict__ bitmap, int64_t nbytes)
{ const int64_t ngroups = total >> 3, npairs = ngroups >> 1;          // pairs of groups; an odd last group is done by the tail thread
  const int64_t nthreads = (int64_t)gridDim.x*blockDim.x;
  const int lane = lane_id();
  const uint4 *vp = reinterpret_cast<const uint4 *>(prof);
  const int64_t wavesz = (int64_t)WAVE*UNROLL;
  const int64_t nwaves = nthreads/WAVE;
  const int64_t wid = ((int64_t)blockIdx.x*blockDim.x+threadIdx.x)/WAVE;
  for (int64_t base = wid*wavesz; base < npairs; base += nwaves*wavesz)
    { uint4 va[UNROLL], vb[UNROLL];
      int64_t g[UNROLL];
#pragma unroll
      for (int u = 0; u < UNROLL; u++)
        { g[u] = base+(int64_t)u*WAVE+lane;
          const bool live = g[u] < npairs;
          va[u] = live ? nt_load(&vp[2*g[u]])   : make_uint4(0,0,0,0);
          vb[u] = live ? nt_load(&vp[2*g[u]+1]) : make_uint4(0,0,0,0);
        }
      unsigned carry = 0;
      if (lane == 0)
        carry = (base > 0) ? prof[base*16-1] : (va[0].x & 0xffff);
#pragma unroll
      for (int u = 0; u < UNROLL; u++)
        { const bool live = g[u] < npairs;
          const unsigned before = (unsigned)wave_from_below((int)vb[u].w,(int)(carry << 16));
          carry = (unsigned)wave_of_last((int)vb[u].w) >> 16;
          const unsigned lo = group_bits(va[u],before,rep), hi = group_bits(vb[u],va[u].w,rep);
          if (live)
            __builtin_nontemporal_store((unsigned short)(lo | (hi << 8)),reinterpret_cast<unsigned short *>(bitmap+2*g[u]));
        }
    }
  if (blockIdx.x == 0 && threadIdx.x == 0)
    { for (int64_t gq = npairs*2; gq < ngroups; gq++)                  // an odd last full group
        { unsigned bits = 0;
          for (int k = 0; k < 8; k++)
            { const int64_t p = gq*8+k;
              if (p > 0)
                { unsigned a = prof[p-1], b = prof[p];
                  unsigned mn = a < b ? a : b, df = a < b ? b-a : a-b;
                  if (mn < (unsigned)rep && df >= 3) bits |= 1u << k;
                }
            }
          bitmap[gq] = (uint8_t)bits;
        }
      int64_t p0 = ngroups << 3;
      if (p0 < total)
        { unsigned bits = 0;
          for (int64_t p = p0; p < total; p++)
            if (p > 0)
              { unsigned a = prof[p-1], b = prof[p];
                unsigned mn = a < b ? a : b, df = a < b ? b-a : a-b;
                if (mn < (unsigned)rep && df >= 3) bits |= 1u << (p-p0);
              }
          bitmap[ngroups] = (uint8_t)bits;
          p0 += 8;
        }
      for (int64_t q = p0 >> 3; q < nbytes; q++) bitmap[q] = 0;
    }
}

// V6 / V7: the flag bytes go through the wave's own LDS block and leave as FULL-WIDTH stores: a wave owns ROWS consecutive
// 1-KB rows per turn (ROWS/UNROLL steps of UNROLL rows in flight), every lane writes its byte of each row to LDS, and after
// the last step lane l reads bytes [W*l, W*l+W) of the wave's ROWS*64 bitmap bytes (W = ROWS/16*4... = ROWS*64/64 = ROWS)
// and stores them with one instruction: ROWS = 4 -> one dword per lane (256 B per wave), 16 -> one dwordx4 (1 KB).
template <int ROWS, int UNROLL, int PLAINST = 0>
__global__ void __launch_bounds__(256)
k_scan_lds(const uint16_t *__restrict__ prof, int64_t total, int rep, uint8_t *__restrict__ bitmap, int64_t nbytes)
{ static_assert(ROWS == 4 || ROWS == 8 || ROWS == 16,"a lane stores 4, 8 or 16 bytes");
  __shared__ __attribute__((aligned(16))) uint8_t s_all[4][ROWS*WAVE];
  const int64_t ngroups = total >> 3;
  const int64_t nthreads = (int64_t)gridDim.x*blockDim.x;
  const int lane = lane_id();
  uint8_t *sb = s_all[threadIdx.x/WAVE];
  const uint4 *vp = reinterpret_cast<const uint4 *>(prof);
  const int64_t wavesz = (int64_t)WAVE*ROWS;
  const int64_t nwaves = nthreads/WAVE;
  const int64_t wid = ((int64_t)blockIdx.x*blockDim.x+threadIdx.x)/WAVE;
  for (int64_t base = wid*wavesz; base < ngroups; base += nwaves*wavesz)
    { unsigned carry = 0;
      if (lane == 0)
        carry = (base > 0) ? prof[base*8-1] : (prof[0] & 0xffff);
#pragma unroll
      for (int st = 0; st < ROWS/UNROLL; st++)
        { uint4 v[UNROLL];
#pragma unroll
          for (int u = 0; u < UNROLL; u++)
            { const int64_t g = base+(int64_t)(st*UNROLL+u)*WAVE+lane;
              v[u] = (g < ngroups) ? nt_load(&vp[g]) : make_uint4(0,0,0,0);
            }
#pragma unroll
          for (int u = 0; u < UNROLL; u++)
            { const unsigned before = (unsigned)wave_from_below((int)v[u].w,(int)(carry << 16));
              carry = (unsigned)wave_of_last((int)v[u].w) >> 16;
              sb[(st*UNROLL+u)*WAVE+lane] = (uint8_t)group_bits(v[u],before,rep);
            }
        }
      // (one wave: its LDS writes are ordered before its reads by the wait the compiler puts in front of the read)
      __builtin_amdgcn_fence(__ATOMIC_RELEASE,"wavefront");
      __builtin_amdgcn_wave_barrier();
      const int64_t ob = base+(int64_t)lane*ROWS;                     // first bitmap byte of this lane's piece
      if (ob+ROWS <= ngroups)
        { if (ROWS == 4)
            __builtin_nontemporal_store(*reinterpret_cast<const unsigned *>(sb+lane*4),reinterpret_cast<unsigned *>(bitmap+ob));
          else if (ROWS == 8)
            { typedef unsigned u2v __attribute__((ext_vector_type(2)));
              __builtin_nontemporal_store(*reinterpret_cast<const u2v *>(sb+lane*8),reinterpret_cast<u2v *>(bitmap+ob));
            }
          else if (PLAINST)
            *reinterpret_cast<cp_u4v *>(bitmap+ob) = *reinterpret_cast<const cp_u4v *>(sb+lane*16);
          else
            __builtin_nontemporal_store(*reinterpret_cast<const cp_u4v *>(sb+lane*16),reinterpret_cast<cp_u4v *>(bitmap+ob));
        }
      else
        for (int k = 0; k < ROWS; k++) if (ob+k < ngroups) bitmap[ob+k] = sb[lane*ROWS+k];
      __builtin_amdgcn_wave_barrier();
    }
  if (blockIdx.x == 0 && threadIdx.x == 0)
    { int64_t p0 = ngroups << 3;
      if (p0 < total)
        { unsigned bits = 0;
          for (int64_t p = p0; p < total; p++)
            if (p > 0)
              { unsigned a = prof[p-1], b = prof[p];
                unsigned mn = a < b ? a : b, df = a < b ? b-a : a-b;
                if (mn < (unsigned)rep && df >= 3) bits |= 1u << (p-p0);
              }
          bitmap[ngroups] = (uint8_t)bits;
          p0 += 8;
        }
      for (int64_t q = p0 >> 3; q < nbytes; q++) bitmap[q] = 0;
    }
}

__global__ void k_fill(uint16_t *p, int64_t n, unsigned seed)
{ for (int64_t i = (int64_t)blockIdx.x*blockDim.x+threadIdx.x; i < n; i += (int64_t)gridDim.x*blockDim.x)
    { unsigned x = (unsigned)(i*2654435761u) ^ seed; x ^= x >> 13; x *= 0x5bd1e995u; x ^= x >> 15;
      // mostly a level around 40 with noise, now and then a jump: like a profile
      unsigned lvl = (((unsigned)(i >> 9) * 2246822519u) >> 28) * 4 + 10;
      // dense = 0: like a profile (noise of +-1, a jump every ~120 positions: ~1.5 % of the positions are candidates);
      // dense = 1: noise of 0..3 (a third of the positions are candidates)
      p[i] = (uint16_t)(lvl + (seed & 1 ? (x & 3) : (x & 1)) + ((x >> 8) % 120 == 0 ? 30 : 0));
    }
}

template <class F> float time_ms(F f, int iters)
{ hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  f(); (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0,0);
  for (int i = 0; i < iters; i++) f();
  (void)hipEventRecord(e1,0); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms,e0,e1);
  return ms/iters;
}

int main(int argc, char **argv)
{ const int64_t total = argc > 1 ? atoll(argv[1]) : 1996136157ll;     // positions of one bench launch
  int rep = 71;
  const int64_t nbytes = (total/64+2)*8;
  uint16_t *prof; uint8_t *bm0, *bm1;
  if (hipMalloc(&prof,total*2+64) != hipSuccess || hipMalloc(&bm0,nbytes) != hipSuccess || hipMalloc(&bm1,nbytes) != hipSuccess) return 1;
  const unsigned dense = argc > 2 ? atoi(argv[2]) : 0;
  hipLaunchKernelGGL(k_fill,dim3(4096),dim3(256),0,0,prof,total,12344u+dense);
  (void)hipDeviceSynchronize();
  std::vector<uint8_t> h0(nbytes), h1(nbytes);
  auto check = [&](const char *name)
    { (void)hipMemcpy(h1.data(),bm1,nbytes,hipMemcpyDeviceToHost);
      const bool same = memcmp(h0.data(),h1.data(),nbytes) == 0;
      printf("      %s bitmap %s V0's\n",name,same ? "==" : "DIFFERS FROM");
      (void)hipMemset(bm1,0xEE,nbytes);
    };
  const double gb = total*2/1e9;
#define RUN(name,kern,UN,blocks,out,iters) \
  { float ms = time_ms([&] { hipLaunchKernelGGL(kern,dim3(blocks),dim3(256),0,0,prof,total,rep,out,nbytes); },iters); \
    printf("%-44s blocks %5d: %8.1f us  %7.1f GB/s (2 B/position)  frac of 8 TB/s %.3f\n",name,blocks,ms*1e3,gb/ms*1e3,gb/ms*1e3/8000.); }
  RUN("V0 product form, unroll 4",(k_scan<0,4>),4,2048,bm0,20);
  (void)hipMemcpy(h0.data(),bm0,nbytes,hipMemcpyDeviceToHost);
  (void)hipMemset(bm1,0xEE,nbytes);
  RUN("V1 no stores",(k_scan<1,4>),4,2048,bm1,20);
  RUN("V2 loads only",(k_scan<2,4>),4,2048,bm1,20);
  (void)hipMemset(bm1,0xEE,nbytes);
  RUN("V3 dword stores by DPP",(k_scan<3,4>),4,2048,bm1,20); check("V3");
  for (int blocks : { 1024, 4096, 8192, 16384 })
    { RUN("V0 product form, unroll 4",(k_scan<0,4>),4,blocks,bm1,10); }
  check("V0/16384");
  RUN("V0 product form, unroll 8",(k_scan<0,8>),8,2048,bm1,10); check("V0/u8");
  RUN("V0 product form, unroll 2",(k_scan<0,2>),2,4096,bm1,10); check("V0/u2");
  for (int blocks : { 2048, 4096, 8192 })
    { RUN("V5 two groups per lane, unroll 2",(k_scan2<2>),2,blocks,bm1,10); }
  check("V5/u2");
  for (int blocks : { 2048, 4096, 8192 })
    { RUN("V5 two groups per lane, unroll 4",(k_scan2<4>),4,blocks,bm1,10); }
  check("V5/u4");
  RUN("V10 a quarter of the byte stores",(k_scan<6,4>),4,2048,bm1,10);
  for (int div : { 1, 4, 16, 64, 256, 1024 })
    { rep = 71 | (div << 16);          // (the kernels compare 16-bit counts with (unsigned short)rep: the high half is only read by MODE 8)
      char nm[64]; snprintf(nm,sizeof nm,"V12 one row in %d stored",div);
      RUN(nm,(k_scan<8,4>),4,2048,bm1,10);
    }
  rep = 71;
  { long nz = 0; for (int64_t q = 0; q < (total >> 3); q++) nz += h0[q] != 0;
    printf("      non-zero bitmap bytes: %.1f %%\n",100.*nz/(total >> 3)); }
  (void)hipMemset(bm1,0,nbytes);
  RUN("V11 only the non-zero bytes stored (bitmap zero before)",(k_scan<7,4>),4,2048,bm1,10); check("V11");
  (void)hipMemset(bm1,0,nbytes);
  RUN("V11 only the non-zero bytes stored (bitmap zero before)",(k_scan<7,4>),4,4096,bm1,10); check("V11/4096");
  { float ms = time_ms([&] { (void)hipMemsetAsync(bm1,0,nbytes,0); },10);
    printf("hipMemsetAsync of the bitmap (%.0f MB): %.1f us\n",nbytes/1e6,ms*1e3); }
  for (int blocks : { 2048, 4096 })
    { RUN("V8 nt loads, PLAIN byte stores",(k_scan<5,4>),4,blocks,bm1,10); }
  check("V8");
  for (int blocks : { 2048, 4096, 8192 })
    { RUN("V9 LDS 16 rows, PLAIN dwordx4 stores",(k_scan_lds<16,4,1>),4,blocks,bm1,10); }
  check("V9");
  for (int blocks : { 2048, 4096, 8192 })
    { RUN("V6 LDS transpose, 4 rows -> dword stores",(k_scan_lds<4,4>),4,blocks,bm1,10); }
  check("V6");
  for (int blocks : { 1024, 2048, 4096, 8192 })
    { RUN("V7 LDS, 8 rows (2 x 4) -> dwordx2 stores",(k_scan_lds<8,4>),4,blocks,bm1,10); }
  check("V7/8");
  for (int blocks : { 1024, 2048, 4096, 8192 })
    { RUN("V7 LDS, 16 rows (4 x 4) -> dwordx4 stores",(k_scan_lds<16,4>),4,blocks,bm1,10); }
  check("V7/16");
  for (int blocks : { 1024, 2048, 4096 })
    { RUN("V7 LDS, 16 rows (2 x 8) -> dwordx4 stores",(k_scan_lds<16,8>),8,blocks,bm1,10); }
  check("V7/16u8");
  return 0;
}

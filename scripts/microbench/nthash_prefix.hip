// Microbenchmark + device check of DESIGN 8.7's next form of the seed kernel's marks: the canonical ntHash (nthash.h:181-235;
// seed.c:28-55 stores it mod 2^31-1) of every k-mer of a stretch of bases
//   (A1) by the K-step fold on the ALU                      (cp_seed.h: cp_kmer_hash, the product's path for K > 64),
//   (A2) by K look-ups each way in an LDS table of pre-rotated seeds (the product's form in sw_mark_all, without its pipelining),
//   (B)  from two prefix-XOR arrays: a wave takes 64 k-mers = 64+K-1 bases in two chunks of 64; lane l rotates its base's seed by
//        -t / +t (t = its position in the tile), a 64-bit XOR scan over the wave (six shuffle steps), P and Q to LDS, then
//        fh(i) = srol^(i+K-1)(P(i+K) ^ P(i)),  rh(i) = srol^-i(Q(i+K) ^ Q(i))   (scripts/proto/nthash_prefix.py has the derivation).
// All three must agree on every k-mer (compared on the host; the first positions also against the literal fold on the CPU).
//   hipcc --offload-arch=gfx950 -O3 -I classpro_amd/csrc scripts/microbench/nthash_prefix.hip -o build_diag/nthash_prefix
//   build_diag/nthash_prefix [K=40] [Mbases=64] [S=0] [const]      (const: S and K as compile-time constants in Bs, for 16/40 and 64/40)
// S > 0 (8, 16, 32 or 64): what the marks really hash -- SHORT taken segments of S k-mers each (about 16 in the seed bench), spread over
// the sequence.  (A2s) a lane per k-mer, all segments' k-mers side by side, as sw_mark_all does; (Bs) a wave takes 64/S segments: their
// base ranges [b, b+S+K-1) side by side, a lane per BASE, scanned straight across the segment borders -- P(p+K) ^ P(p) cancels whatever
// came before p, so no segmented scan is needed, only the K bases of a k-mer next to each other -- then a lane per k-mer.  A segment of S
// k-mers costs S+K-1 base terms: the prefix form pays for the K-1 bases the k-mers of a short segment share.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#include "cp_seed.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr,"%s: %s\n",#x,hipGetErrorString(e_)); exit(2); } } while (0)
#define WAVE 64
#define KMAXT 64                                                   // K <= 64 for A2 and B (two chunks of 64 bases cover 64+K-1)

__device__ __host__ inline uint64_t srol_n(uint64_t v, int m)      // srol^m, any integer m: the low 33 and the high 31 bits rotate on their own
{ const uint64_t LO = (1ull << 33)-1, HI = (1ull << 31)-1;
  int a = m % 33, b = m % 31;
  if (a < 0) a += 33;
  if (b < 0) b += 31;
  uint64_t lo = v & LO, hi = v >> 33;
  lo = ((lo << a) | (lo >> (33-a))) & LO;
  hi = ((hi << b) | (hi >> (31-b))) & HI;
  return lo | (hi << 33);
}

__global__ void k_fold(const char *seq, int64_t n, int K, int *out)
{ const int64_t i = (int64_t)blockIdx.x*blockDim.x+threadIdx.x;
  if (i+K <= n) out[i] = cp_kmer_hash(seq+i,0,K);
}

__device__ inline int code_of(uint64_t s)                          // 0..3 = A C G T, 4 = no seed
{ return s == 0x3c8bfbb395c60474ull ? 0 : s == 0x3193c18562a02b4cull ? 1 : s == 0x20323ed082572324ull ? 2 : s == 0x295549f54be24456ull ? 3 : 4; }

__global__ void __launch_bounds__(256) k_table(const char *seq, int64_t n, int K, int *out)
{ __shared__ uint64_t rot[5][KMAXT+1];
  __shared__ uint8_t cf[256], cr[256];
  const uint64_t sd[5] = { 0x3c8bfbb395c60474ull, 0x3193c18562a02b4cull, 0x20323ed082572324ull, 0x295549f54be24456ull, 0 };
  for (int q = threadIdx.x; q < 5*KMAXT; q += blockDim.x) rot[q/KMAXT][q%KMAXT] = srol_n(sd[q/KMAXT],q%KMAXT);
  for (int q = threadIdx.x; q < 256; q += blockDim.x) { cf[q] = (uint8_t)code_of(cp_nt_seed((unsigned)q)); cr[q] = (uint8_t)code_of(cp_nt_seed_rc((unsigned)q)); }
  __syncthreads();
  const int64_t i = (int64_t)blockIdx.x*blockDim.x+threadIdx.x;
  if (i+K > n) return;
  uint64_t fh = 0, rh = 0;
  for (int t = 0; t < K; t++)
    { const unsigned c = (unsigned char)seq[i+t];
      fh ^= rot[cf[c]][K-1-t];
      rh ^= rot[cr[c]][t];
    }
  out[i] = (int)((rh < fh ? rh : fh) % CP_SEED_MOD);
}

__device__ inline uint64_t xor_scan(uint64_t x, int lane)         // inclusive, over the wave
{
#pragma unroll
  for (int o = 1; o < WAVE; o <<= 1)
    { const uint64_t y = (uint64_t)__shfl_up((unsigned long long)x,o);
      if (lane >= o) x ^= y;
    }
  return x;
}

__global__ void __launch_bounds__(256) k_prefix(const char *seq, int64_t n, int K, int *out)
{ __shared__ uint64_t sP[4][2*WAVE+1], sQ[4][2*WAVE+1];            // P(t), Q(t) for t = 0 .. 128 of the wave's tile
  const int lane = threadIdx.x & (WAVE-1), w = threadIdx.x >> 6;
  const int64_t j0 = ((int64_t)blockIdx.x*4+w)*WAVE;               // first k-mer (= first base) of the tile
  uint64_t cp = 0, cq = 0;
  if (lane == 0) { sP[w][0] = 0; sQ[w][0] = 0; }
#pragma unroll
  for (int c = 0; c < 2; c++)
    { const int t = c*WAVE+lane;
      uint64_t xf = 0, xr = 0;
      if (j0+t < n)
        { const unsigned ch = (unsigned char)seq[j0+t];
          xf = srol_n(cp_nt_seed(ch),-t);
          xr = srol_n(cp_nt_seed_rc(ch),t);
        }
      xf = xor_scan(xf,lane)^cp;
      xr = xor_scan(xr,lane)^cq;
      sP[w][t+1] = xf; sQ[w][t+1] = xr;
      cp = (uint64_t)__shfl((unsigned long long)xf,WAVE-1);
      cq = (uint64_t)__shfl((unsigned long long)xr,WAVE-1);
    }
  __syncthreads();
  const int64_t i = j0+lane;
  if (i+K <= n)
    { const uint64_t fh = srol_n(sP[w][lane+K]^sP[w][lane],lane+K-1);
      const uint64_t rh = srol_n(sQ[w][lane+K]^sQ[w][lane],-lane);
      out[i] = (int)((rh < fh ? rh : fh) % CP_SEED_MOD);
    }
}

// ---- short segments: segment g = the S k-mers starting at g*stride ---------------------------------------------------------------
__global__ void k_fold_seg(const char *seq, int64_t nseg, int64_t stride, int S, int K, int *out)
{ const int64_t q = (int64_t)blockIdx.x*blockDim.x+threadIdx.x;
  if (q < nseg*S) out[q] = cp_kmer_hash(seq+(q/S)*stride+q%S,0,K);
}

__global__ void __launch_bounds__(256) k_table_seg(const char *seq, int64_t nseg, int64_t stride, int S, int K, int *out)
{ __shared__ uint64_t rot[5][KMAXT+1];
  __shared__ uint8_t cf[256], cr[256];
  const uint64_t sd[5] = { 0x3c8bfbb395c60474ull, 0x3193c18562a02b4cull, 0x20323ed082572324ull, 0x295549f54be24456ull, 0 };
  for (int q = threadIdx.x; q < 5*KMAXT; q += blockDim.x) rot[q/KMAXT][q%KMAXT] = srol_n(sd[q/KMAXT],q%KMAXT);
  for (int q = threadIdx.x; q < 256; q += blockDim.x) { cf[q] = (uint8_t)code_of(cp_nt_seed((unsigned)q)); cr[q] = (uint8_t)code_of(cp_nt_seed_rc((unsigned)q)); }
  __syncthreads();
  const int64_t q = (int64_t)blockIdx.x*blockDim.x+threadIdx.x;
  if (q >= nseg*S) return;
  const char *km = seq+(q/S)*stride+q%S;
  uint64_t fh = 0, rh = 0;
  for (int t = 0; t < K; t++)
    { const unsigned c = (unsigned char)km[t];
      fh ^= rot[cf[c]][K-1-t];
      rh ^= rot[cr[c]][t];
    }
  out[q] = (int)((rh < fh ? rh : fh) % CP_SEED_MOD);
}

#define MAXB (WAVE+8*(KMAXT-1))                                    // bases of 64/S segments, S >= 8, K <= 64
template <int CS, int CK>                                          // CS, CK > 0: S and K known to the compiler (divisions by constants); 0: run-time values
__global__ void __launch_bounds__(256) k_prefix_seg(const char *seq, int64_t nseg, int64_t stride, int S_, int K_, int *out)
{ __shared__ uint64_t sP[4][MAXB+1], sQ[4][MAXB+1];
  const int S = CS ? CS : S_, K = CK ? CK : K_;
  const int lane = threadIdx.x & (WAVE-1), w = threadIdx.x >> 6;
  const int per = WAVE/S, span = S+K-1, nbase = per*span;          // segments per wave, bases per segment, bases per wave
  const int64_t g0 = ((int64_t)blockIdx.x*4+w)*per;                // the wave's first segment
  uint64_t cp = 0, cq = 0;
  if (lane == 0) { sP[w][0] = 0; sQ[w][0] = 0; }
  for (int c0 = 0; c0 < nbase; c0 += WAVE)
    { const int u = c0+lane;
      uint64_t xf = 0, xr = 0;
      if (u < nbase && g0+u/span < nseg)
        { const unsigned ch = (unsigned char)seq[(g0+u/span)*stride+u%span];
          xf = srol_n(cp_nt_seed(ch),-u);
          xr = srol_n(cp_nt_seed_rc(ch),u);
        }
      xf = xor_scan(xf,lane)^cp;
      xr = xor_scan(xr,lane)^cq;
      if (u < nbase) { sP[w][u+1] = xf; sQ[w][u+1] = xr; }
      cp = (uint64_t)__shfl((unsigned long long)xf,WAVE-1);
      cq = (uint64_t)__shfl((unsigned long long)xr,WAVE-1);
    }
  __syncthreads();
  const int sg = lane/S, i = lane%S, p = sg*span+i;                // my k-mer: segment, position in it, position among the wave's bases
  if (g0+sg < nseg)
    { const uint64_t fh = srol_n(sP[w][p+K]^sP[w][p],p+K-1);
      const uint64_t rh = srol_n(sQ[w][p+K]^sQ[w][p],-p);
      out[(g0+sg)*S+i] = (int)((rh < fh ? rh : fh) % CP_SEED_MOD);
    }
}

int main(int argc, char **argv)
{ const int K = argc > 1 ? atoi(argv[1]) : 40;
  const int64_t n = (int64_t)(argc > 2 ? atoi(argv[2]) : 64) << 20;
  const int S = argc > 3 ? atoi(argv[3]) : 0;
  if (K < 1 || K > KMAXT || n < K || (S != 0 && S != 8 && S != 16 && S != 32 && S != 64)) { fprintf(stderr,"K in 1..64, S in 0 8 16 32 64\n"); return 2; }
  const int64_t stride = 4*S+K, nseg = S ? (n-K-S)/stride : 0;
  const int64_t nk = S ? nseg*S : n-K+1;
  std::vector<char> h((size_t)n);
  uint64_t r = 88172645463325252ull;
  const char *L = "ACGTacgtNnUuRYKM";
  for (int64_t i = 0; i < n; i++)
    { r ^= r << 13; r ^= r >> 7; r ^= r << 17;
      h[(size_t)i] = (r & 63) == 0 ? L[(r >> 8) & 15] : "ACGT"[(r >> 8) & 3];
    }
  char *d_seq; int *d_o[3];
  CK(hipMalloc(&d_seq,(size_t)n));
  CK(hipMemcpy(d_seq,h.data(),(size_t)n,hipMemcpyHostToDevice));
  for (int v = 0; v < 3; v++) { CK(hipMalloc(&d_o[v],(size_t)nk*4)); CK(hipMemset(d_o[v],0xff,(size_t)nk*4)); }
  const unsigned nb = (unsigned)((nk+255)/256);
  const char *name[3] = { "A1 fold on the ALU (cp_kmer_hash)", "A2 table of rotated seeds in LDS  ", S ? (argc > 4 ? "Bs prefix-XOR, S and K constants  " : "Bs prefix-XOR over the base ranges") : "B  two prefix-XOR scans per tile  " };
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float ms[3];
  for (int v = 0; v < 3; v++)
    { for (int rep = 0; rep < 6; rep++)
        { if (rep == 1) CK(hipEventRecord(e0,0));
          if (S)
            { if (v == 0) hipLaunchKernelGGL(k_fold_seg,dim3(nb),dim3(256),0,0,d_seq,nseg,stride,S,K,d_o[0]);
              if (v == 1) hipLaunchKernelGGL(k_table_seg,dim3(nb),dim3(256),0,0,d_seq,nseg,stride,S,K,d_o[1]);
              if (v == 2)
                { if (S == 16 && K == 40 && argc > 4) hipLaunchKernelGGL((k_prefix_seg<16,40>),dim3(nb),dim3(256),0,0,d_seq,nseg,stride,S,K,d_o[2]);
                  else if (S == 64 && K == 40 && argc > 4) hipLaunchKernelGGL((k_prefix_seg<64,40>),dim3(nb),dim3(256),0,0,d_seq,nseg,stride,S,K,d_o[2]);
                  else hipLaunchKernelGGL((k_prefix_seg<0,0>),dim3(nb),dim3(256),0,0,d_seq,nseg,stride,S,K,d_o[2]);
                }
              continue;
            }
          if (v == 0) hipLaunchKernelGGL(k_fold,dim3(nb),dim3(256),0,0,d_seq,n,K,d_o[0]);
          if (v == 1) hipLaunchKernelGGL(k_table,dim3(nb),dim3(256),0,0,d_seq,n,K,d_o[1]);
          if (v == 2) hipLaunchKernelGGL(k_prefix,dim3(nb),dim3(256),0,0,d_seq,n,K,d_o[2]);
        }
      CK(hipEventRecord(e1,0)); CK(hipEventSynchronize(e1)); CK(hipGetLastError());
      CK(hipEventElapsedTime(&ms[v],e0,e1)); ms[v] /= 5;
    }
  std::vector<int> o[3];
  for (int v = 0; v < 3; v++) { o[v].resize((size_t)nk); CK(hipMemcpy(o[v].data(),d_o[v],(size_t)nk*4,hipMemcpyDeviceToHost)); }
  int64_t bad12 = 0, bad13 = 0, badcpu = 0;
  for (int64_t i = 0; i < nk; i++) { bad12 += o[0][(size_t)i] != o[1][(size_t)i]; bad13 += o[0][(size_t)i] != o[2][(size_t)i]; }
  const int64_t ncpu = nk < 2000000 ? nk : 2000000;
  for (int64_t i = 0; i < ncpu; i++) badcpu += cp_kmer_hash(h.data()+(S ? (i/S)*stride+i%S : i),0,K) != o[2][(size_t)i];
  if (S) printf("ntHash of the k-mers of %lld segments of %d k-mers (%d bases each), K = %d, %lld k-mers (1/64 of the letters odd):\n",
                (long long)nseg,S,S+K-1,K,(long long)nk);
  else printf("ntHash of every k-mer, K = %d, %lld k-mers (1/64 of the letters odd):\n",K,(long long)nk);
  for (int v = 0; v < 3; v++) printf("  %s  %8.3f ms  %7.1f G k-mers/s\n",name[v],ms[v],nk/ms[v]/1e6);
  printf("  differences: A2 vs A1 %lld, B vs A1 %lld, B vs the literal fold on the CPU (first %lld) %lld\n",
         (long long)bad12,(long long)bad13,(long long)ncpu,(long long)badcpu);
  return bad12 || bad13 || badcpu ? 1 : 0;
}

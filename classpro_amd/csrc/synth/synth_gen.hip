// synth_gen.hip -- device-side synthetic HiFi read-set generator (TEST / BENCH INFRASTRUCTURE, not the product).
//
// SURVEY.md section 8(d) asks for a seeded *profile synthesiser* that can make the 8-Gbase (200 Mbp x 40x)
// and larger workloads in seconds: FastK, the MHC download and an exact k-mer counter are not available
// offline, and the numpy generator (classpro_amd/synth.py, exact classes by hashing) does ~20 Mbases/s.
// Everything here is a pure function of (seed, position) or (seed, read id, position), so any read range can
// be generated on its own (multi-GPU shards regenerate exactly the reads they own) and nothing is tiled.
//
//   genome     haplotype A: uniform ACGT; one micro-satellite / homopolymer run per 16-kb slot; most slots hold
//              a copy of one of the segmental-repeat families (3 kb, 3-5 copies each);
//              haplotype B = A with SNPs at rate `het`, none inside the homozygous slots
//   reads      genomic span ~ N(r,(0.2 r)^2) clipped to [min_len,max_len], random haplotype / strand / start;
//              errors per genomic position of the span: substitution, 1-base deletion, 1-base insertion
//              (duplicate), the indels `hp_mult` times likelier inside a homopolymer
//   profile    count of a clean read k-mer (no edited base, no junction) = number of clean read k-mers, over all
//              reads and both strands, on the same genomic window or on a window with the same sequence (the
//              other haplotype when the window holds no SNP, the other copies of its repeat family); a k-mer
//              touched by a read error has count 1.  O(bases); the windows' sequence classes are known by
//              construction instead of being found by hashing 2G k-mers.
//   truth      the "relative profile" of prof2class.c (genomic multiplicity of the k-mer: 0 E, 1 H, 2 D, >=3 R).
//
// Phases (host orchestration in classpro_amd/synth_dev.py; prefix sums are torch.cumsum between kernels):
//   sg_genome -> [cumsum of SNP flags] -> sg_reads_pass1 (read lengths, +-1 coverage differences per window)
//   -> [cumsum: clean coverage per window and haplotype] -> sg_famtot -> sg_totals (+ FASTK-style histogram)
//   -> sg_reads_pass2 (bases, counts, truth for any read range).
// The C entry points take device pointers and a hipStream_t; no torch types.
#include <hip/hip_runtime.h>
#include <stdint.h>

#define SG_WAVE 64
#define SG_SLOT_SHIFT 14                 // 16-kb slots
#define SG_SLOT (1 << SG_SLOT_SHIFT)

struct sg_params
  { uint64_t seed;
    int64_t  G;                          // haplotype length
    int64_t  n_reads;
    int32_t  K;
    int32_t  copy_off, copy_len;         // a slot's repeat copy = [copy_off, copy_off+copy_len) inside the slot
    int32_t  lc_span;                    // the slot's low-complexity run starts inside [0, lc_span)
    int32_t  homo_every;                 // every homo_every-th slot carries no SNP (0 = none)
    uint32_t t_het, t_sub, t_del, t_ins; // rates as thresholds on 32 random bits
    int32_t  hp_mult;
    int32_t  read_mean, read_sd, min_len, max_len;
  };

__host__ __device__ __forceinline__ uint64_t sg_mix(uint64_t x)
{ x ^= x >> 30; x *= 0xbf58476d1ce4e5b9ULL;
  x ^= x >> 27; x *= 0x94d049bb133111ebULL;
  x ^= x >> 31;
  return x;
}
#define SG_C1 0x9E3779B97F4A7C15ULL
#define SG_C2 0xD1B54A32D192ED03ULL
#define SG_TAG_GENOME 0x1111ULL
#define SG_TAG_TEMPL  0x2222ULL
#define SG_TAG_LC     0x3333ULL
#define SG_TAG_SNP    0x4444ULL
#define SG_TAG_READ   0x5555ULL
#define SG_TAG_EVENT  0x6666ULL

__device__ __forceinline__ uint64_t sg_key(uint64_t seed, uint64_t tag) { return sg_mix(seed*SG_C1+tag); }

// ---------------------------------------------------------------------------------------------------------
//  genome: gen[p] = A | B << 2 | snp << 4
// ---------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_sg_genome(sg_params P, const int32_t *__restrict__ slot_fam, uint8_t *__restrict__ gen)
{ const uint64_t kg = sg_key(P.seed,SG_TAG_GENOME), kt = sg_key(P.seed,SG_TAG_TEMPL),
                 kl = sg_key(P.seed,SG_TAG_LC), ks = sg_key(P.seed,SG_TAG_SNP);
  const int64_t nfull = P.G >> SG_SLOT_SHIFT;
  for (int64_t p = (int64_t)blockIdx.x*blockDim.x+threadIdx.x; p < P.G; p += (int64_t)gridDim.x*blockDim.x)
    { const int64_t slot = p >> SG_SLOT_SHIFT;
      const int off = (int)(p & (SG_SLOT-1));
      const int f = (slot < nfull) ? slot_fam[slot] : -1;
      unsigned a;
      const uint64_t bh = sg_mix(kl ^ ((uint64_t)slot*SG_C2));
      const int q = (int)(bh % (uint64_t)P.lc_span);
      const int ulen = 1+(int)((bh >> 20) % 3);
      const int reps = 6+(int)((bh >> 24) % (ulen == 3 ? 7 : 10));
      if (f >= 0 && off >= P.copy_off && off < P.copy_off+P.copy_len)
        a = (unsigned)(sg_mix(kt ^ ((uint64_t)(f >> 3)*SG_C1) ^ ((uint64_t)(off-P.copy_off)*SG_C2)) >> 13) & 3u;
      else if (off >= q && off < q+ulen*reps)
        a = (unsigned)(sg_mix(kl ^ ((uint64_t)slot*SG_C1) ^ ((uint64_t)((off-q) % ulen)+1)) >> 17) & 3u;
      else
        a = (unsigned)(sg_mix(kg ^ ((uint64_t)p*SG_C2)) >> 11) & 3u;
      const uint64_t sh = sg_mix(ks ^ ((uint64_t)p*SG_C2));
      const bool homo = P.homo_every > 0 && (slot % P.homo_every) == (P.homo_every-1);
      const bool snp = !homo && (uint32_t)sh < P.t_het;
      const unsigned b = snp ? ((a+1+(unsigned)((sh >> 32) % 3)) & 3u) : a;
      gen[p] = (uint8_t)(a | (b << 2) | (snp ? 16u : 0u));
    }
}

// ---------------------------------------------------------------------------------------------------------
//  One read, one wave: walks the read's genomic span 64 positions at a time and hands every position to
//  `emit`.  Output index o of a position's first output base, and for the k-mer that ENDS at that base:
//  clean <=> it starts at an output index i >= 0 with no dirty base in [i,o] and no junction inside.
// ---------------------------------------------------------------------------------------------------------
enum { SG_NONE = 0, SG_SUB = 1, SG_DEL = 2, SG_INS = 3 };

struct sg_read { int hap, strand; int64_t s; int L; uint64_t ek; };

__device__ __forceinline__ sg_read sg_read_attr(const sg_params &P, int64_t id)
{ sg_read R;
  const uint64_t h = sg_mix(sg_key(P.seed,SG_TAG_READ) ^ ((uint64_t)id*SG_C2));
  const uint64_t h2 = sg_mix(h ^ SG_C1);
  R.hap = (int)(h & 1); R.strand = (int)((h >> 1) & 1);
  // Box-Muller on two 24-bit uniforms
  const float u1 = ((float)((h >> 8) & 0xffffff)+1.f)*(1.f/16777217.f), u2 = (float)((h >> 32) & 0xffffff)*(1.f/16777216.f);
  const float z = sqrtf(-2.f*logf(u1))*cosf(6.28318530718f*u2);
  int64_t L = (int64_t)lrintf((float)P.read_mean+(float)P.read_sd*z);
  int64_t mx = P.max_len-256;                  // insertions lengthen the read; keep it below max_len
  if (mx > P.G) mx = P.G;
  if (L < P.min_len) L = P.min_len;
  if (L > mx) L = mx;
  R.L = (int)L;
  R.s = (int64_t)(h2 % (uint64_t)(P.G-L+1));
  R.ek = sg_mix(sg_key(P.seed,SG_TAG_EVENT) ^ ((uint64_t)id*SG_C1));
  return R;
}

__device__ __forceinline__ int sg_wave_excl_add(int x, int *total)
{ const int lane = threadIdx.x & 63;
  int inc = x;
  for (int o = 1; o < SG_WAVE; o <<= 1)
    { int y = __shfl_up(inc,o); if (lane >= o) inc += y; }
  *total = __shfl(inc,SG_WAVE-1);
  return inc-x;
}
__device__ __forceinline__ int sg_wave_excl_max(int x, int *total)       // values >= -1
{ const int lane = threadIdx.x & 63;
  int inc = x;
  for (int o = 1; o < SG_WAVE; o <<= 1)
    { int y = __shfl_up(inc,o); if (lane >= o && y > inc) inc = y; }
  *total = __shfl(inc,SG_WAVE-1);
  int ex = __shfl_up(inc,1);
  return lane == 0 ? -1 : ex;
}

struct sg_pos                                      // what `emit` sees for one genomic position
  { bool valid; int64_t p; int ev; unsigned base;  // base: the read's base at p before/after a substitution (after)
    int o;                                         // output index of the first base this position emits
    bool clean;                                    // the k-mer ending at output o is clean (ev NONE or INS only)
  };

template <class Emit>
__device__ __forceinline__ int sg_walk_read(const sg_params &P, const uint8_t *__restrict__ gen, const sg_read &R, Emit &emit)
{ const int lane = threadIdx.x & 63;
  const int Km1 = P.K-1;
  int nout = 0, last_dirty = -1, last_junc = -1;
  for (int c = 0; c < R.L; c += SG_WAVE)
    { sg_pos q;
      q.valid = c+lane < R.L;
      q.p = R.s+c+lane;
      const unsigned gb = q.valid ? gen[q.p] : 0u;
      unsigned base = R.hap ? ((gb >> 2) & 3u) : (gb & 3u);
      unsigned prev = __shfl_up(base,1);
      if (lane == 0)
        { prev = 4u;
          if (q.p > 0) { const unsigned g1 = gen[q.p-1]; prev = R.hap ? ((g1 >> 2) & 3u) : (g1 & 3u); }
        }
      const uint64_t u = sg_mix(R.ek ^ ((uint64_t)q.p*SG_C2));
      const uint32_t lo = (uint32_t)u;
      const uint32_t m = (base == prev) ? (uint32_t)P.hp_mult : 1u;
      const uint32_t td = P.t_sub+P.t_del*m, ti = td+P.t_ins*m;
      q.ev = !q.valid ? SG_NONE : lo < P.t_sub ? SG_SUB : lo < td ? SG_DEL : lo < ti ? SG_INS : SG_NONE;
      if (q.ev == SG_SUB) base = (base+1+(unsigned)((u >> 32) % 3)) & 3u;
      q.base = base;
      const int n = !q.valid ? 0 : q.ev == SG_DEL ? 0 : q.ev == SG_INS ? 2 : 1;
      int tot;
      q.o = nout+sg_wave_excl_add(n,&tot);
      const int myd = q.ev == SG_SUB ? q.o : q.ev == SG_INS ? q.o+1 : -1;
      const int myj = q.ev == SG_DEL ? q.o : -1;      // junction between outputs o-1 and o
      int dmax, jmax;
      int dex = sg_wave_excl_max(myd,&dmax), jex = sg_wave_excl_max(myj,&jmax);
      if (last_dirty > dex) dex = last_dirty;
      if (last_junc > jex) jex = last_junc;
      const int i = q.o-Km1;
      q.clean = q.valid && (q.ev == SG_NONE || q.ev == SG_INS) && i >= 0 && dex < i && jex <= i;
      emit(q);
      nout += tot;
      if (dmax > last_dirty) last_dirty = dmax;
      if (jmax > last_junc) last_junc = jmax;
    }
  return nout;
}

// ---- pass 1: read lengths + coverage differences ------------------------------------------------------
struct sg_emit_cov
  { int32_t *diff; int Km1; bool prev_clean; int nclean; int64_t last_p;
    __device__ __forceinline__ void operator()(const sg_pos &q)
    { const int lane = threadIdx.x & 63;
      const uint64_t cm = __ballot(q.clean);
      const bool before = lane == 0 ? prev_clean : (((cm >> (lane-1)) & 1) != 0);
      if (q.valid && q.clean != before)
        atomicAdd(&diff[q.p-Km1],q.clean ? 1 : -1);
      const uint64_t vm = __ballot(q.valid);
      const int hi = 63-__builtin_clzll(vm);                   // last valid lane of the chunk
      prev_clean = ((cm >> hi) & 1) != 0;
      last_p = __shfl(q.p,hi);
      nclean += __popcll(cm);
    }
  };

__global__ void __launch_bounds__(SG_WAVE)
k_sg_pass1(sg_params P, const uint8_t *__restrict__ gen, int32_t *__restrict__ diffA, int32_t *__restrict__ diffB,
           int32_t *__restrict__ rlen, unsigned long long *__restrict__ nerr)
{ const int64_t id = blockIdx.x;
  if (id >= P.n_reads) return;
  const sg_read R = sg_read_attr(P,id);
  sg_emit_cov E;
  E.diff = R.hap ? diffB : diffA; E.Km1 = P.K-1; E.prev_clean = false; E.nclean = 0; E.last_p = R.s;
  const int nout = sg_walk_read(P,gen,R,E);
  if ((threadIdx.x & 63) == 0)
    { if (E.prev_clean) atomicAdd(&E.diff[E.last_p+1-E.Km1],-1);
      rlen[id] = nout;
      const int plen = nout-E.Km1;
      if (plen > E.nclean) atomicAdd(nerr,(unsigned long long)(plen-E.nclean));
    }
}

// ---- family totals ---------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_sg_famtot(sg_params P, const int32_t *__restrict__ snpcum, const int32_t *__restrict__ cntA, const int32_t *__restrict__ cntB,
            const int32_t *__restrict__ fam_off, const int32_t *__restrict__ fam_slots, int n_fam,
            int32_t *__restrict__ famtot, uint8_t *__restrict__ fammult)
{ const int nwin = P.copy_len-P.K+1;
  const int64_t n = (int64_t)n_fam*nwin;
  for (int64_t t = (int64_t)blockIdx.x*blockDim.x+threadIdx.x; t < n; t += (int64_t)gridDim.x*blockDim.x)
    { const int f = (int)(t/nwin), o = (int)(t%nwin);
      int64_t sum = 0; int mult = 0;
      for (int c = fam_off[f]; c < fam_off[f+1]; c++)
        { const int64_t g = ((int64_t)fam_slots[c] << SG_SLOT_SHIFT)+P.copy_off+o;
          const bool het = snpcum[g+P.K] != snpcum[g];
          sum += cntA[g]+(het ? 0 : cntB[g]);
          mult += het ? 1 : 2;
        }
      famtot[t] = (int32_t)(sum > 0x7fffffff ? 0x7fffffff : sum);
      fammult[t] = (uint8_t)(mult > 255 ? 255 : mult);
    }
}

// ---- per-window totals, truth and the FASTK-style histogram of distinct k-mers ----------------------
#define SG_LHIST 512
__global__ void __launch_bounds__(256)
k_sg_totals(sg_params P, const int32_t *__restrict__ snpcum, const int32_t *__restrict__ cntA, const int32_t *__restrict__ cntB,
            const int32_t *__restrict__ slot_fam, const int32_t *__restrict__ famtot, const uint8_t *__restrict__ fammult,
            uint16_t *__restrict__ totA, uint16_t *__restrict__ totB, uint8_t *__restrict__ relA, uint8_t *__restrict__ relB,
            unsigned long long *__restrict__ hist)
{ __shared__ unsigned lh[SG_LHIST];
  for (int k = threadIdx.x; k < SG_LHIST; k += blockDim.x) lh[k] = 0;
  __syncthreads();
  const int64_t nwin_g = P.G-P.K+1, nfull = P.G >> SG_SLOT_SHIFT;
  const int nwin = P.copy_len-P.K+1;
  for (int64_t g = (int64_t)blockIdx.x*blockDim.x+threadIdx.x; g < P.G; g += (int64_t)gridDim.x*blockDim.x)
    { if (g >= nwin_g) { totA[g] = totB[g] = 0; relA[g] = relB[g] = 0; continue; }
      const bool het = snpcum[g+P.K] != snpcum[g];
      const int64_t slot = g >> SG_SLOT_SHIFT;
      const int off = (int)(g & (SG_SLOT-1));
      const int f = (slot < nfull) ? slot_fam[slot] : -1;
      const bool incopy = f >= 0 && off >= P.copy_off && off+P.K <= P.copy_off+P.copy_len;
      int64_t ta; int ra; bool rep_a = true;
      if (incopy)
        { const int64_t t = (int64_t)(f >> 3)*nwin+(off-P.copy_off);
          ta = famtot[t]; ra = fammult[t];
          rep_a = (f & 7) == 0;                       // the family's first copy stands for the class in the histogram
        }
      else
        { ta = (int64_t)cntA[g]+(het ? 0 : cntB[g]); ra = het ? 1 : 2; }
      const int64_t tb = het ? (int64_t)cntB[g] : ta;
      const int rb = het ? 1 : ra;
      const int ca = (int)(ta > 32767 ? 32767 : ta), cb = (int)(tb > 32767 ? 32767 : tb);
      totA[g] = (uint16_t)ca; totB[g] = (uint16_t)cb;
      relA[g] = (uint8_t)ra; relB[g] = (uint8_t)rb;
      if (rep_a && ca > 0) { if (ca < SG_LHIST) atomicAdd(&lh[ca],1u); else atomicAdd(&hist[ca],1ull); }
      if (het && cb > 0)   { if (cb < SG_LHIST) atomicAdd(&lh[cb],1u); else atomicAdd(&hist[cb],1ull); }
    }
  __syncthreads();
  for (int k = threadIdx.x; k < SG_LHIST; k += blockDim.x)
    if (lh[k]) atomicAdd(&hist[k],(unsigned long long)lh[k]);
}

// ---- pass 2: bases, counts, truth of reads [first, first+count) -----------------------------------------
struct sg_emit_out
  { const uint16_t *tot; const uint8_t *rel;
    char *seq; uint16_t *prof; uint8_t *truth;
    int rlen, plen, Km1, strand;
    __device__ __forceinline__ void put_base(int o, unsigned b) const
    { const unsigned acgt = 0x54474341u;                       // 'A' 'C' 'G' 'T', one byte each
      if (strand) seq[rlen-1-o] = (char)(acgt >> (8u*(3u-b))); else seq[o] = (char)(acgt >> (8u*b));
    }
    __device__ __forceinline__ void put_kmer(int o, bool clean, int64_t g) const      // the k-mer ending at output o
    { const int i = o-Km1;
      if (i < 0) return;
      const int j = strand ? plen-1-i : i;
      prof[j] = clean ? tot[g] : (uint16_t)1;
      if (truth) truth[j] = clean ? rel[g] : (uint8_t)0;
    }
    __device__ __forceinline__ void operator()(const sg_pos &q) const
    { if (!q.valid || q.ev == SG_DEL) return;
      put_base(q.o,q.base);
      put_kmer(q.o,q.clean,q.p-Km1);
      if (q.ev == SG_INS)
        { put_base(q.o+1,q.base);
          put_kmer(q.o+1,false,0);
        }
    }
  };

__global__ void __launch_bounds__(SG_WAVE)
k_sg_pass2(sg_params P, const uint8_t *__restrict__ gen, const uint16_t *__restrict__ totA, const uint16_t *__restrict__ totB,
           const uint8_t *__restrict__ relA, const uint8_t *__restrict__ relB, int64_t first, int count,
           const int64_t *__restrict__ seq_off, char *__restrict__ seq, uint16_t *__restrict__ prof, uint8_t *__restrict__ truth,
           int32_t *__restrict__ err)
{ const int r = blockIdx.x;
  if (r >= count) return;
  const sg_read R = sg_read_attr(P,first+r);
  sg_emit_out E;
  E.Km1 = P.K-1;
  E.rlen = (int)(seq_off[r+1]-seq_off[r]); E.plen = E.rlen-E.Km1; E.strand = R.strand;
  E.tot = R.hap ? totB : totA; E.rel = R.hap ? relB : relA;
  const int64_t po = seq_off[r]-(int64_t)r*E.Km1;
  E.seq = seq+seq_off[r]; E.prof = prof+po; E.truth = truth ? truth+po : nullptr;
  const int nout = sg_walk_read(P,gen,R,E);
  if ((threadIdx.x & 63) == 0 && nout != E.rlen) atomicOr(err,1);
}

// ---------------------------------------------------------------------------------------------------------
static thread_local char sg_err[256];
extern "C" const char *sg_last_error(void) { return sg_err; }
#define SGCHK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { snprintf(sg_err,sizeof(sg_err),"%s: %s",#call,hipGetErrorString(e_)); return -1; } } while (0)
#include <stdio.h>

static int sg_grid(int64_t n) { int64_t b = (n+255)/256; if (b > 256*32) b = 256*32; if (b < 1) b = 1; return (int)b; }

extern "C" int sg_genome(const sg_params *P, const int32_t *d_slot_fam, uint8_t *d_gen, void *stream)
{ hipLaunchKernelGGL(k_sg_genome,dim3(sg_grid(P->G)),dim3(256),0,(hipStream_t)stream,*P,d_slot_fam,d_gen);
  SGCHK(hipGetLastError());
  return 0;
}

extern "C" int sg_reads_pass1(const sg_params *P, const uint8_t *d_gen, int32_t *d_diffA, int32_t *d_diffB, int32_t *d_rlen,
                              unsigned long long *d_nerr, void *stream)
{ if (P->n_reads <= 0) return 0;
  hipLaunchKernelGGL(k_sg_pass1,dim3((unsigned)P->n_reads),dim3(SG_WAVE),0,(hipStream_t)stream,*P,d_gen,d_diffA,d_diffB,d_rlen,d_nerr);
  SGCHK(hipGetLastError());
  return 0;
}

extern "C" int sg_famtot(const sg_params *P, const int32_t *d_snpcum, const int32_t *d_cntA, const int32_t *d_cntB,
                         const int32_t *d_fam_off, const int32_t *d_fam_slots, int n_fam, int32_t *d_famtot, uint8_t *d_fammult,
                         void *stream)
{ if (n_fam <= 0) return 0;
  const int64_t n = (int64_t)n_fam*(P->copy_len-P->K+1);
  hipLaunchKernelGGL(k_sg_famtot,dim3(sg_grid(n)),dim3(256),0,(hipStream_t)stream,*P,d_snpcum,d_cntA,d_cntB,d_fam_off,d_fam_slots,n_fam,
                     d_famtot,d_fammult);
  SGCHK(hipGetLastError());
  return 0;
}

extern "C" int sg_totals(const sg_params *P, const int32_t *d_snpcum, const int32_t *d_cntA, const int32_t *d_cntB,
                         const int32_t *d_slot_fam, const int32_t *d_famtot, const uint8_t *d_fammult,
                         uint16_t *d_totA, uint16_t *d_totB, uint8_t *d_relA, uint8_t *d_relB, unsigned long long *d_hist, void *stream)
{ hipLaunchKernelGGL(k_sg_totals,dim3(sg_grid(P->G)),dim3(256),0,(hipStream_t)stream,*P,d_snpcum,d_cntA,d_cntB,d_slot_fam,d_famtot,d_fammult,
                     d_totA,d_totB,d_relA,d_relB,d_hist);
  SGCHK(hipGetLastError());
  return 0;
}

extern "C" int sg_reads_pass2(const sg_params *P, const uint8_t *d_gen, const uint16_t *d_totA, const uint16_t *d_totB,
                              const uint8_t *d_relA, const uint8_t *d_relB, int64_t first, int count, const int64_t *d_seq_off,
                              char *d_seq, uint16_t *d_prof, uint8_t *d_truth, int32_t *d_err, void *stream)
{ if (count <= 0) return 0;
  hipLaunchKernelGGL(k_sg_pass2,dim3((unsigned)count),dim3(SG_WAVE),0,(hipStream_t)stream,*P,d_gen,d_totA,d_totB,d_relA,d_relB,first,count,
                     d_seq_off,d_seq,d_prof,d_truth,d_err);
  SGCHK(hipGetLastError());
  return 0;
}

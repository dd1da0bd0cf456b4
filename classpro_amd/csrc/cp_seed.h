// cp_seed.h -- scalar pieces of the `-s` seed path (src/seed.c) shared by the device kernel (cp_seed_wave.h, which
// holds the algorithm) and the host-side unit tests: the canonical ntHash of a k-mer and the interval overlap test.
#pragma once
#include <stdint.h>
#include "cp_types.h"

#ifndef CP_HDM
#ifdef __HIPCC__
#define CP_HDM __host__ __device__ __forceinline__
#else
#define CP_HDM inline
#endif
#endif
#ifdef __HIPCC__
#define CP_HDN __host__ __device__ __attribute__((noinline))
#else
#define CP_HDN inline
#endif

#define CP_SEED_W      1000        // seed.c:23 WSIZE
#define CP_SEED_W_REP  200         // seed.c:24 WSIZE_REP
#define CP_SEED_MOD    2147483647  // seed.c:26

// ---- canonical ntHash of the k-mer starting at read position j, mod 2^31-1 (seed.c:28-55) -----------------
CP_HDM uint64_t cp_nt_seed(unsigned c)                             // seedTab, nthash.h:26-59: A C G T/U in both cases (and the codes 1..7); else 0
{ switch (c)
    { case 'A': case 'a': case 4: case 5: return 0x3c8bfbb395c60474ull;
      case 'C': case 'c': case 7:         return 0x3193c18562a02b4cull;
      case 'G': case 'g': case 3:         return 0x20323ed082572324ull;
      case 'T': case 't': case 'U': case 'u': case 1: return 0x295549f54be24456ull;
    }
  return 0;
}
CP_HDM uint64_t cp_nt_seed_rc(unsigned c)                          // seedTab[c & cpOff], nthash.h:17,226-235: the complement by the low three bits
{ switch (c & 7u)
    { case 1: return 0x295549f54be24456ull;                        // ..001  A a        -> T
      case 3: return 0x20323ed082572324ull;                        // ..011  C c        -> G
      case 4: case 5: return 0x3c8bfbb395c60474ull;                // ..10x  T t U u    -> A
      case 7: return 0x3193c18562a02b4cull;                        // ..111  G g        -> C
    }
  return 0;
}
CP_HDM uint64_t cp_nt_srol(uint64_t v)                             // rol1 + swapbits033, nthash.h:181-207
{ v = (v << 1) | (v >> 63);
  const uint64_t x = (v ^ (v >> 33)) & 1;
  return v ^ (x | (x << 33));
}
// the fold of nthash.h:215-235 (NTF64 / NTR64): equal to the reference's rolling values
CP_HDN int cp_kmer_hash(const char *seq, int j, int K)
{ uint64_t fh = 0, rh = 0;
  for (int i = 0; i < K; i++)
    { fh = cp_nt_srol(fh) ^ cp_nt_seed((unsigned char)seq[j+i]);
      rh = cp_nt_srol(rh) ^ cp_nt_seed_rc((unsigned char)seq[j+K-1-i]);
    }
  return (int)((rh < fh ? rh : fh) % CP_SEED_MOD);
}

CP_HDM bool cp_seed_ovlp(int ab, int ae, int bb, int be)             // seed.c:120-127
{ const int lo = ab > bb ? ab : bb, hi = (ae < be ? ae : be)-1; return lo <= hi; }

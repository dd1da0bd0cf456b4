// classpro_main.cpp -- `ClassPro`-compatible command line on top of the C ABI (libclasspro_amd.so).
//
// Same process contract as the reference binary (src/const.c:14-17, src/ClassPro.c:348-631):
//   ClassPro [-vs] [-T<int(4)>] [-c<int>] [-r<int(20000)>] [-P<tmp_dir(./)>] [-N<fastk_root>]
//            [-M<model_path>] <source>[.db|.dam|.f[ast][aq][.gz]]
// inputs  <fk_root>.hist, <fk_root>.prof, <dir>/.<root>.pidx.N, <dir>/.<root>.prof.N   (FASTK)
// output  <dir>/<root>.class : "@name comment\nseq\n+\nlabels\n" per read              (ClassPro.c:289)
//         for a Dazzler database also the tracks .<root>.class.anno/.data (and .<root>.rep.*)
//
// The reference's parallel structure -- T pthreads over contiguous read ranges, one input handle per thread,
// per-thread temp files concatenated in order by merge_files (ClassPro.c:530,558-614; io.c:70-112) -- becomes:
//   * ONE input sharded over all visible devices (CLASSPRO_DEVICES=0,1,.. or =0,0 to put two shards on one
//     GPU): the input text is taken in windows; a window's reads are split into one contiguous range per device,
//     balanced by bases; a range goes through its device in batches of up to 256 Mbases;
//   * per device one feeder thread (hipSetDevice, its own cp_params, three batch slots = stream + cp_workspace +
//     pinned and device buffers) and one completion thread;
//   * -T host threads (a pool) do every data-parallel host loop: indexing the input text (fastx_index.h: plain
//     FASTA in parallel chunks, FASTQ by one thread, .gz inflated by one thread = the .gz ceiling), staging copies
//     into pinned memory, formatting the records;
//   * no temp files and no merge pass: the size of every record is known once its window is indexed, so the window's
//     stretch of <root>.class is fallocated and mapped at once and the pool formats every record at its final offset
//     -- the ordered concatenation merge_files produces, without the second copy.
// FASTK profile code strings are shipped as stored (0.27 B/base) and decoded on the device (cp_decode_profiles);
// database bases are shipped 2-bit packed (cp_unpack_bases).
// -s (seed.c, Dazzler inputs only, as in the reference): cp_find_seeds_batch after the classification; the .class.data
// track then carries the seed labels (ClassPro.c:293) and .rep.anno/.data the repeat-mask intervals (seed.c:531-566).
#include <hip/hip_runtime.h>
#include <zlib.h>
#include <dirent.h>
#include <fcntl.h>
#include <unistd.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <strings.h>
#include <cstdio>
#include <cstdarg>
#include <cstdlib>
#include <cstring>
#include <cstdint>
#include <string>
#include <vector>
#include <deque>
#include <chrono>
#include <thread>
#include <mutex>
#include <atomic>
#include <memory>
#include <condition_variable>
#include <csignal>
#include <cerrno>
#include "../../../include/classpro_amd.h"
#include "host_io.h"
#include "dazz_db.h"
#include "thread_pool.h"
#include "fastx_index.h"

static void remember_mapping(const void *m, size_t len);      // (SIGBUS handling, below)

static const char *USAGE = "[-vs] [-T<int(4)>] [-c<int>] [-r<int(20000)>] "
                           "[-P<tmp_dir(./)>] [-N<fastk_root>] [-M<model_path>] "
                           "<source>[.db|.dam|.f[ast][aq][.gz]";                      // const.c:14-17
static const char *EXT[10] = { ".db", ".dam", ".fastq", ".fasta", ".fq", ".fa",
                               ".fastq.gz", ".fasta.gz", ".fq.gz", ".fa.gz" };          // ClassPro.h:326-330

#define HIPOK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) die("%s: %s: %s\n",PROG,#call,hipGetErrorString(e_)); } while (0)
#define CPOK(call)  do { if ((call) < 0) die("%s\n",cp_last_error()); } while (0)

typedef std::chrono::steady_clock::time_point tp_t;
static tp_t now() { return std::chrono::steady_clock::now(); }
static double secs(tp_t a, tp_t b) { return std::chrono::duration<double>(b-a).count(); }

// ---- read-only memory map of a file --------------------------------------------------------------------
struct MapFile
  { const char *p = nullptr; size_t len = 0; int fd = -1;
    bool open(const std::string &path)
    { fd = ::open(path.c_str(),O_RDONLY);
      if (fd < 0) return false;
      struct stat st;
      if (fstat(fd,&st) != 0) return false;
      len = (size_t)st.st_size;
      if (len > 0)
        { void *m = mmap(nullptr,len,PROT_READ,MAP_PRIVATE,fd,0);
          if (m == MAP_FAILED) return false;
          p = (const char *)m;
          madvise(m,len,MADV_SEQUENTIAL);
        }
      return true;
    }
    ~MapFile() { if (p) munmap((void *)p,len); if (fd >= 0) close(fd); }
  };

// ---- FASTK profile index over memory-mapped parts (libfastk.c:1267-1370, 1414-1465) ----------------------
struct ProfileMap
  { int kmer = 0, nparts = 0;
    int64_t nreads = 0;
    std::vector<int64_t> index;         // end offset of read i inside its part
    std::vector<int64_t> nbase;         // reads before the end of part p
    std::vector<std::unique_ptr<MapFile>> part;
    bool open(const std::string &fk_root)
    { std::string dir = path_to(fk_root), root = root_of(fk_root,".prof");
      FILE *f = fopen((dir+"/"+root+".prof").c_str(),"rb");
      if (!f) return false;
      int smer, nthreads;
      if (fread(&smer,4,1,f) != 1 || fread(&nthreads,4,1,f) != 1) { fclose(f); return false; }
      fclose(f);
      const std::string prefix = dir+"/."+root+".";
      for (int p = 0; p < nthreads; p++)
        { std::string nm = prefix+"pidx."+std::to_string(p+1);
          FILE *g = fopen(nm.c_str(),"rb");
          if (!g) die("Profile part %s is misssing ?\n",nm.c_str());
          int k; int64_t first, n;
          if (fread(&k,4,1,g) != 1 || fread(&first,8,1,g) != 1 || fread(&n,8,1,g) != 1) die("Profile part %s is truncated\n",nm.c_str());
          if (k != smer) die("Profile part %s does not have k-mer length matching stub ?\n",nm.c_str());
          size_t o = index.size();
          index.resize(o+(size_t)n);
          if (n > 0 && fread(index.data()+o,8,(size_t)n,g) != (size_t)n) die("Profile part %s is truncated\n",nm.c_str());
          fclose(g);
          nreads += n;
          nbase.push_back(nreads);
          part.emplace_back(new MapFile());
          nm = prefix+"prof."+std::to_string(p+1);
          if (!part.back()->open(nm)) die("Profile part %s is misssing ?\n",nm.c_str());
        }
      kmer = smer; nparts = nthreads;
      return true;
    }
    int part_of(int64_t id) const { int w = 0; while (w < nparts && id >= nbase[w]) w++; return w; }
    int64_t code_len(int64_t id) const
    { if (id >= nreads) return 0;
      const int w = part_of(id);
      const int64_t first = (w == 0) ? 0 : nbase[w-1];
      return index[(size_t)id]-((id == first) ? 0 : index[(size_t)id-1]);
    }
    // code string of read `id` whose part is w (w = part_of(id))
    void fetch(int64_t id, int w, const uint8_t **code, int64_t *len) const
    { const int64_t first = (w == 0) ? 0 : nbase[w-1];
      const int64_t off = (id == first) ? 0 : index[(size_t)id-1];
      const int64_t end = index[(size_t)id];
      if (end < off || (size_t)end > part[w]->len) die("Profile part %d is truncated\n",w+1);
      *code = (const uint8_t *)part[w]->p+off;
      *len = end-off;
    }
  };

// ---- blocking queue ---------------------------------------------------------------------------------------
template <class T> struct Chan
  { std::deque<T> q; std::mutex m; std::condition_variable cv; bool closed = false;
    void push(T v) { { std::lock_guard<std::mutex> lk(m); q.push_back(std::move(v)); } cv.notify_one(); }
    void close() { { std::lock_guard<std::mutex> lk(m); closed = true; } cv.notify_all(); }
    bool pop(T &v)
    { std::unique_lock<std::mutex> lk(m);
      cv.wait(lk,[&] { return closed || !q.empty(); });
      if (q.empty()) return false;
      v = std::move(q.front()); q.pop_front();
      return true;
    }
  };

// ---- a window of the input: text + records, alive until its last batch has been written --------------------
struct Window
  { std::vector<char> owned;                   // .gz: inflated text
    std::vector<FxRec> recs;                   // FASTX: parsed records; DB: name = header, seq = packed bases
    std::deque<std::string> keep;              // inherited comment / database headers
    int64_t first_id = 0;                      // input index of recs[0]
    std::vector<int64_t> out_off;              // offset of every record in <root>.class (+ end)
    std::vector<int64_t> trk_off;              // database: offset of every read in .class.data (+ end)
    // The window's stretch of the output files, mapped shared.  The output's pages are what the host pays most for:
    // a fresh page-cache / tmpfs page costs the kernel an allocation, a clear and a memcg charge, and ONE file takes
    // them no faster from many writers than from one (write(2) serialises on the inode lock: 5-8 GB/s; first-touch
    // faults of a shared mapping: 3.6-5.6 GB/s).  fallocate() hands a file its pages at 14.8 GB/s from one thread, and
    // pages that exist are filled through a mapping at 130 GB/s by 16 threads (scripts/microbench/tmpfs_write.cpp,
    // tmpfs_falloc.cpp; profiles/r02_tmpfs_write.txt).  So: an allocator thread fallocates a window's stretch as soon
    // as the window is indexed -- while HIP starts and the device works -- and the formatter threads write the
    // records in place.
    char *omap = nullptr, *tmap = nullptr; size_t omap_len = 0, tmap_len = 0; int64_t omap_base = 0, tmap_base = 0;
    int64_t out_lo = 0, out_hi = 0, trk_lo = 0, trk_hi = 0;
    bool allocated = false;                    // set by the allocator thread (under Run::am); formatters wait for it
    char *out_at(int64_t off) const { return omap+(off-omap_base); }
    char *trk_at(int64_t off) const { return tmap+(off-tmap_base); }
    void map_out(int fd, int64_t lo, int64_t hi, bool track)
    { if (hi <= lo) return;
      const int64_t base = lo & ~(int64_t)4095;
      if (ftruncate(fd,(off_t)hi) != 0) die("%s: cannot size the output\n",PROG);
      int mfd = fd; off_t moff = (off_t)base;
      // diagnostic knob (tests): the k-th output mapping of the run is backed by an EMPTY anonymous file, so the first
      // store into it faults with SIGBUS exactly as a page that the file system cannot allocate does
      static std::atomic<int> n_maps{0};
      if (const char *e = getenv("CLASSPRO_DEBUG_ENOSPC_MAPPING"))
        if (!track && atoi(e) == n_maps.fetch_add(1))
          { mfd = memfd_create("classpro_enospc",0); moff = 0;
            if (mfd < 0) die("%s: memfd_create failed\n",PROG);
          }
      void *m = mmap(nullptr,(size_t)(hi-base),PROT_READ|PROT_WRITE,MAP_SHARED,mfd,moff);
      if (m == MAP_FAILED) die("%s: cannot map the output file\n",PROG);
      remember_mapping(m,(size_t)(hi-base));
      if (track) { tmap = (char *)m; tmap_len = (size_t)(hi-base); tmap_base = base; trk_lo = lo; trk_hi = hi; }
      else       { omap = (char *)m; omap_len = (size_t)(hi-base); omap_base = base; out_lo = lo; out_hi = hi; }
    }
    // (the mappings are left to process exit: munmap of a gigabyte holds the address-space lock against every
    //  thread's page faults, and the process leaves through _exit right after the last window)
  };

struct BatchJob { std::shared_ptr<Window> w; size_t r0, r1; };

// ---- one batch slot of a device ------------------------------------------------------------------------------
struct Slot
  { char *h_seq = nullptr, *h_lab = nullptr, *h_seed = nullptr, *d_seed = nullptr; uint8_t *h_code = nullptr, *h_pack = nullptr;
    int64_t *h_soff = nullptr, *h_poff = nullptr, *h_coff = nullptr, *h_boff = nullptr;
    char *d_seq = nullptr, *d_lab = nullptr; uint8_t *d_code = nullptr, *d_pack = nullptr; uint16_t *d_prof = nullptr;
    int64_t *d_soff = nullptr, *d_poff = nullptr, *d_coff = nullptr, *d_boff = nullptr;
    size_t cap_bases = 0, cap_reads = 0, cap_code = 0;
    hipStream_t st = nullptr;
    cp_workspace *ws = nullptr;
    // the batch in the slot
    BatchJob job;
    int n = 0; int64_t bases = 0, kmers = 0, codes = 0, packed = 0;
    std::vector<int32_t> slot_of;              // record (relative to r0) -> index among classified reads, -1 = short read
    std::vector<uint32_t> rec_of;              // classified read -> record (relative to r0)
    void alloc(size_t bases_cap, size_t reads_cap, bool with_pack, bool with_seeds)
    { if (with_seeds) { HIPOK(hipHostMalloc((void **)&h_seed,bases_cap)); HIPOK(hipMalloc((void **)&d_seed,bases_cap)); }
      cap_bases = bases_cap; cap_reads = reads_cap; cap_code = bases_cap;
      HIPOK(hipHostMalloc((void **)&h_seq,bases_cap)); HIPOK(hipHostMalloc((void **)&h_lab,bases_cap));
      HIPOK(hipHostMalloc((void **)&h_code,cap_code));
      HIPOK(hipHostMalloc((void **)&h_soff,(reads_cap+1)*8)); HIPOK(hipHostMalloc((void **)&h_poff,(reads_cap+1)*8));
      HIPOK(hipHostMalloc((void **)&h_coff,(reads_cap+1)*8));
      if (with_pack)
        { HIPOK(hipHostMalloc((void **)&h_pack,bases_cap/4+reads_cap)); HIPOK(hipMalloc((void **)&d_pack,bases_cap/4+reads_cap));
          HIPOK(hipHostMalloc((void **)&h_boff,(reads_cap+1)*8)); HIPOK(hipMalloc((void **)&d_boff,(reads_cap+1)*8));
        }
      HIPOK(hipMalloc((void **)&d_seq,bases_cap)); HIPOK(hipMalloc((void **)&d_lab,bases_cap));
      HIPOK(hipMalloc((void **)&d_code,cap_code)); HIPOK(hipMalloc((void **)&d_prof,bases_cap*2));
      HIPOK(hipMalloc((void **)&d_soff,(reads_cap+1)*8)); HIPOK(hipMalloc((void **)&d_poff,(reads_cap+1)*8));
      HIPOK(hipMalloc((void **)&d_coff,(reads_cap+1)*8));
      HIPOK(hipStreamCreate(&st));
      CPOK(cp_workspace_create(&ws));
    }
  };

// ---- everything the pipeline threads share -----------------------------------------------------------------
struct Run
  { int K = 40, Km1 = 39;
    bool is_db = false, is_dam = false, verbose = false, seeds = false;
    std::vector<std::vector<int32_t>> rep;      // -s: repeat-mask intervals of every read (b,e pairs), filled batch by batch
    int rlen_opt = 20000, hcov = 0, dcov = 0, max_rlen = CP_MAX_READ_LEN;
    std::string model_path;
    ProfileMap P;
    ThreadPool *pool = nullptr;
    int out_fd = -1, trk_fd = -1;
    std::atomic<int64_t> total_bases{0};
    std::atomic<int64_t> t_stage_us{0}, t_wait_us{0}, t_write_us{0};
    size_t batch_bases = (size_t)256 << 20, batch_reads = 1 << 17;
    std::mutex am; std::condition_variable acv;   // a window's pages are there before anyone formats into them: a formatter
                                                  // that runs ahead of the allocator would take pages one fault at a time and
                                                  // slow the allocator down on the same file
    int64_t pre_upto = 0;                         // (under am) the output has its pages up to here from the up-front allocation
    bool pre_wait_all = false; bool pre_done = true;
    // window throttle
    std::mutex wm; std::condition_variable wcv; int windows_alive = 0;
  };

[[maybe_unused]] static void pwrite_all(int fd, const char *buf, size_t n, int64_t off)
{ while (n > 0)
    { ssize_t w = pwrite(fd,buf,n,(off_t)off);
      if (w <= 0) die("%s: write to the output failed\n",PROG);
      buf += w; n -= (size_t)w; off += w;
    }
}

// The outputs are written through shared mappings of files that were sized with ftruncate: a page that could not be
// allocated (a full tmpfs / disk, a quota) shows up as SIGBUS at the first store into it, not as a failed write().  The
// reference's fprintf path reports "no space" and exits 1; so does this one: fallocate's result is checked (give_pages)
// and, for file systems without fallocate, a SIGBUS handler says what happened.  Either way the partial outputs go.
static char g_out_files[4][1024];
static void remember_output(const std::string &p)
{ for (auto &f : g_out_files) if (!f[0]) { snprintf(f,sizeof(f),"%s",p.c_str()); return; } }
// The mapped ranges of the outputs (map_out registers them): a SIGBUS is "no space" only when it comes from a store into
// one of them.  From anywhere else -- a truncated or changed mapped INPUT, a hardware error -- the default action is
// restored and the signal raised again, so the diagnosis and the core are the real ones and no output is deleted on a guess.
// The registry is a push-only linked list behind an atomic head: any number of mappings (a window adds one or two and a
// human-scale input has hundreds of windows), appended by the indexing thread, walked by the handler without a lock
// (a lock-free atomic pointer load is async-signal-safe; nodes are never freed or changed once published).
struct OutMap { uintptr_t lo, hi; OutMap *next; };
static std::atomic<OutMap *> g_out_maps{nullptr};
static void remember_mapping(const void *m, size_t len)
{ OutMap *n = new OutMap{(uintptr_t)m,(uintptr_t)m+len,g_out_maps.load(std::memory_order_relaxed)};
  while (!g_out_maps.compare_exchange_weak(n->next,n,std::memory_order_release,std::memory_order_relaxed)) { }
}
static void on_sigbus(int sig, siginfo_t *si, void *)
{ const uintptr_t a = (uintptr_t)si->si_addr;
  bool ours = false;
  for (const OutMap *m = g_out_maps.load(std::memory_order_acquire); m; m = m->next) if (a >= m->lo && a < m->hi) ours = true;
  if (!ours)
    { signal(sig,SIG_DFL);
      raise(sig);
      return;
    }
  static const char msg[] = "ClassPro: no space left on the output device (a page of the mapped output could not be allocated)\n";
  ssize_t w = write(2,msg,sizeof(msg)-1); (void)w;
  for (auto &f : g_out_files) if (f[0]) unlink(f);
  _exit(1);
}
static void give_pages(int fd, int mode, int64_t lo, int64_t n, const char *what)
{ if (n <= 0) return;
  int rc;
  do rc = fallocate(fd,mode,(off_t)lo,(off_t)n); while (rc != 0 && errno == EINTR);
  if (rc == 0 || errno == EOPNOTSUPP || errno == ENOSYS || errno == EINVAL)
    return;                                    // (no fallocate here: pages come at first touch, on_sigbus reports a full device)
  const int e = errno;
  for (auto &f : g_out_files) if (f[0]) unlink(f);
  die("%s: cannot allocate %lld bytes of %s: %s\n",PROG,(long long)n,what,strerror(e));
}

static inline void unpack_bases(const unsigned char *pk, int len, char *dst)      // DB.c:342-381 (Uncompress_Read + Upper_Read)
{ static const char letter[4] = { 'A', 'C', 'G', 'T' };
  for (int k = 0; k < len; k++)
    dst[k] = letter[(pk[k >> 2] >> (6-2*(k & 3))) & 3];
}

// ---- a device: feeder (stage + submit) and completion (wait + format + write) threads ----------------------
struct Device
  { Run *R; int dev; int index;
    static constexpr int NSLOT = 3;
    Slot slot[NSLOT];
    cp_params *params = nullptr;
    Chan<BatchJob> in;
    Chan<int> inflight, freeslots;
    std::thread feeder, completer;

    void start(Run *run, int device, int idx)
    { R = run; dev = device; index = idx;
      feeder = std::thread([this] { feed(); });
      completer = std::thread([this] { complete(); });
    }

    void feed()
    { HIPOK(hipSetDevice(dev));
      CPOK(cp_params_create_model(R->K,R->rlen_opt,R->hcov,R->dcov,R->model_path.empty() ? NULL : R->model_path.c_str(),&params));
      if (index == 0 && R->verbose)                           // ClassPro.c:550, wall.c:169-172, ClassPro.c:572
        { int c4[4];
          cp_params_export(params,c4,NULL,NULL,NULL,NULL,NULL,NULL);
          fprintf(stderr,"    Estimated R-threshold = %d\n",c4[CP_REPEAT]);
          if (R->model_path.empty())
            fprintf(stderr,"Error model not specified. Using the default error model.\n");
          fprintf(stderr,"Classifying %d-mers...\n",R->K);
          size_t tb[3] = {0,0,0};
          cp_params_tables(params,&tb[0],&tb[1],&tb[2]);
          fprintf(stderr,"device tables: logp_trans %zu MB, unrel binomial %zu MB, P(error in) %zu MB%s\n",tb[0] >> 20,tb[1] >> 20,tb[2] >> 20,
                  (tb[0] && tb[1] && tb[2]) ? "" : " (0 = computed on the spot: no device memory for it, or switched off)");
        }
      // the slots (0.8 GB of pinned memory each: the costly part of start-up) come up on a helper thread, one by
      // one, while the first batches are already moving
      std::thread allocator([this]
        { HIPOK(hipSetDevice(dev));
          for (int k = 0; k < NSLOT; k++)
            { slot[k].alloc(R->batch_bases+(size_t)R->max_rlen,R->batch_reads,R->is_db,R->seeds);
              freeslots.push(k);
            }
        });
      allocator.detach();
      BatchJob job;
      while (in.pop(job))
        { int k;
          if (!freeslots.pop(k)) break;
          stage(slot[k],job);
          submit(slot[k]);
          inflight.push(k);
        }
      inflight.close();
    }

    // fill the slot's pinned buffers from the window (host threads of the pool)
    void stage(Slot &s, const BatchJob &job)
    { const tp_t t0 = now();
      s.job = job;
      const Window &w = *job.w;
      const size_t nrec = job.r1-job.r0;
      s.slot_of.assign(nrec,-1);
      s.rec_of.clear();
      s.n = 0; s.bases = s.kmers = s.codes = s.packed = 0;
      s.h_soff[0] = s.h_poff[0] = s.h_coff[0] = 0;
      if (R->is_db) s.h_boff[0] = 0;
      std::vector<const uint8_t *> csrc;
      csrc.reserve(nrec);
      int part = R->P.part_of(w.first_id+(int64_t)job.r0);
      for (size_t q = 0; q < nrec; q++)
        { const FxRec &r = w.recs[job.r0+q];
          const int64_t id = w.first_id+(int64_t)(job.r0+q);
          if ((int)r.rlen <= R->Km1) continue;                  // ClassPro.c:209-226: printed, not classified
          while (part < R->P.nparts && id >= R->P.nbase[part]) part++;
          const uint8_t *code; int64_t clen;
          R->P.fetch(id,part,&code,&clen);
          if (clen > 2*(int64_t)r.rlen+2)
            die("Read %lld: profile code of %lld bytes is longer than any code of a read of %u bases\n",(long long)id+1,(long long)clen,r.rlen);
          s.slot_of[q] = s.n;
          s.rec_of.push_back((uint32_t)q);
          csrc.push_back(code);
          s.bases += r.rlen; s.kmers += (int64_t)r.rlen-R->Km1; s.codes += clen; s.n++;
          s.h_soff[s.n] = s.bases; s.h_poff[s.n] = s.kmers; s.h_coff[s.n] = s.codes;
          if (R->is_db) { s.packed += (r.rlen+3) >> 2; s.h_boff[s.n] = s.packed; }
        }
      if ((size_t)s.codes > s.cap_code || (size_t)s.bases > s.cap_bases || (size_t)s.n > s.cap_reads)
        die("%s: internal error: batch exceeds its slot\n",PROG);
      const int n = s.n;
      const int64_t blk = 64;
      R->pool->parallel_for((n+blk-1)/blk,[&](int64_t t)
        { const int a = (int)(t*blk), b = (int)std::min<int64_t>(n,(t+1)*blk);
          for (int i = a; i < b; i++)
            { const FxRec &r = w.recs[job.r0+s.rec_of[(size_t)i]];
              memcpy(s.h_code+s.h_coff[i],csrc[(size_t)i],(size_t)(s.h_coff[i+1]-s.h_coff[i]));
              if (R->is_db)
                { memcpy(s.h_pack+s.h_boff[i],r.seq,(size_t)(s.h_boff[i+1]-s.h_boff[i]));
                  unpack_bases((const unsigned char *)r.seq,(int)r.rlen,s.h_seq+s.h_soff[i]);     // for the output record
                }
              else
                r.copy_seq(s.h_seq+s.h_soff[i]);
            }
        });
      R->t_stage_us += (int64_t)(secs(t0,now())*1e6);
    }

    void submit(Slot &b)
    { if (b.n == 0) return;
      if (!R->is_db)
        HIPOK(hipMemcpyAsync(b.d_seq,b.h_seq,(size_t)b.bases,hipMemcpyHostToDevice,b.st));
      HIPOK(hipMemcpyAsync(b.d_code,b.h_code,(size_t)b.codes,hipMemcpyHostToDevice,b.st));
      HIPOK(hipMemcpyAsync(b.d_soff,b.h_soff,(size_t)(b.n+1)*8,hipMemcpyHostToDevice,b.st));
      HIPOK(hipMemcpyAsync(b.d_poff,b.h_poff,(size_t)(b.n+1)*8,hipMemcpyHostToDevice,b.st));
      HIPOK(hipMemcpyAsync(b.d_coff,b.h_coff,(size_t)(b.n+1)*8,hipMemcpyHostToDevice,b.st));
      if (R->is_db)                                               // 2-bit bases in, characters made on the device
        { HIPOK(hipMemcpyAsync(b.d_pack,b.h_pack,(size_t)b.packed,hipMemcpyHostToDevice,b.st));
          HIPOK(hipMemcpyAsync(b.d_boff,b.h_boff,(size_t)(b.n+1)*8,hipMemcpyHostToDevice,b.st));
          CPOK(cp_unpack_bases(b.d_pack,b.d_boff,b.d_soff,b.n,b.d_seq,b.st));
        }
      // Fetch_Profile runs on the device: only the code string crosses PCIe.  plen is taken from rlen; the
      // reference's rlen != plen+Km1 check (ClassPro.c:234) is made by the decode kernel (see complete()).
      CPOK(cp_decode_profiles(b.ws,b.d_code,b.d_coff,b.d_poff,b.n,b.d_prof,b.st));
      CPOK(cp_classify_batch(params,b.ws,b.d_seq,b.d_soff,b.d_prof,b.d_poff,b.n,b.bases,b.kmers,b.d_lab,b.st));
      HIPOK(hipMemcpyAsync(b.h_lab,b.d_lab,(size_t)b.bases,hipMemcpyDeviceToHost,b.st));
      if (R->seeds)                                               // ClassPro.c:281-282
        { CPOK(cp_find_seeds_batch(params,b.ws,b.d_seq,b.d_soff,b.d_prof,b.d_poff,b.d_lab,b.n,b.bases,b.kmers,b.d_seed,b.st));
          HIPOK(hipMemcpyAsync(b.h_seed,b.d_seed,(size_t)b.bases,hipMemcpyDeviceToHost,b.st));
        }
    }

    void complete()
    { HIPOK(hipSetDevice(dev));
      int k;
      while (inflight.pop(k))
        { Slot &b = slot[k];
          const tp_t t0 = now();
          if (b.n > 0)
            { HIPOK(hipStreamSynchronize(b.st));
              if (cp_workspace_check(b.ws) != CP_OK)
                { // a failed decode: find the read on the host so the message is the reference's (ClassPro.c:234-237)
                  std::vector<uint16_t> tmp((size_t)R->max_rlen);
                  for (int i = 0; i < b.n; i++)
                    { int rlen = (int)(b.h_soff[i+1]-b.h_soff[i]);
                      int plen = cp_decode_profile(b.h_code+b.h_coff[i],b.h_coff[i+1]-b.h_coff[i],tmp.data(),R->max_rlen);
                      if (plen >= 0 && rlen != plen+R->Km1)
                        die("Read %lld: rlen (%d) != plen+Km1 (%d)\n",(long long)(b.job.w->first_id+(int64_t)b.job.r0+b.rec_of[(size_t)i])+1,rlen,plen+R->Km1);
                    }
                  die("%s\n",cp_last_error());
                }
            }
          if (R->seeds && b.n > 0)                                // the .rep intervals of the batch's reads
            { const int64_t cap = cp_rep_masks_capacity(b.ws);
              std::vector<int32_t> cnt((size_t)b.n), pairs((size_t)(cap > 0 ? cap : 1)*2);
              std::vector<int64_t> off((size_t)b.n+1);
              CPOK(cp_get_rep_masks(b.ws,cnt.data(),off.data(),pairs.data(),cap > 0 ? cap : 1));
              for (int i = 0; i < b.n; i++)
                { const int64_t id = b.job.w->first_id+(int64_t)b.job.r0+b.rec_of[(size_t)i];
                  R->rep[(size_t)id].assign(pairs.begin()+2*off[(size_t)i],pairs.begin()+2*(off[(size_t)i]+cnt[(size_t)i]));
                }
            }
          const tp_t t1 = now();
          write_out(b);
          R->t_wait_us += (int64_t)(secs(t0,t1)*1e6);
          R->t_write_us += (int64_t)(secs(t1,now())*1e6);
          R->total_bases += b.bases;
          b.job.w.reset();
          freeslots.push(k);
        }
    }

    // records of the batch at their final offsets (ClassPro.c:215,289; tracks: ClassPro.c:217-223,290-304)
    void write_out(Slot &b)
    { const Window &w = *b.job.w;
      const size_t r0 = b.job.r0, nrec = b.job.r1-r0;
      { std::unique_lock<std::mutex> lk(R->am);
        R->acv.wait(lk,[&] { return w.allocated; });
      }
      const int64_t blk = 64;
      R->pool->parallel_for(((int64_t)nrec+blk-1)/blk,[&](int64_t t)
        { const size_t a = (size_t)(t*blk), z = std::min(nrec,(size_t)((t+1)*blk));
          char *o = w.out_at(w.out_off[r0+a]);
          for (size_t q = a; q < z; q++)
            { const FxRec &r = w.recs[r0+q];
              *o++ = '@';
              memcpy(o,r.name,r.name_len); o += r.name_len;
              if (r.cmt) { *o++ = ' '; memcpy(o,r.cmt,r.cmt_len); o += r.cmt_len; }
              *o++ = '\n';
              const int i = b.slot_of[q];
              const char *lab = nullptr;
              if (i >= 0)
                { memcpy(o,b.h_seq+b.h_soff[i],r.rlen);
                  lab = b.h_lab+b.h_soff[i];
                }
              else if (R->is_db) unpack_bases((const unsigned char *)r.seq,(int)r.rlen,o);
              else r.copy_seq(o);
              o += r.rlen;
              *o++ = '\n'; *o++ = '+'; *o++ = '\n';
              if (lab) memcpy(o,lab,r.rlen); else memset(o,'N',r.rlen);
              if (R->is_db)                                       // Compress_Read of the state codes, E=0 R=1 H=2 D=3
                { const char *trk = (R->seeds && i >= 0) ? b.h_seed+b.h_soff[i] : o;   // -s: the seed labels (ClassPro.c:293)
                  unsigned char *d = (unsigned char *)w.trk_at(w.trk_off[r0+q]);
                  const uint32_t nby = (r.rlen+3) >> 2;
                  for (uint32_t y = 0; y < nby; y++)
                    { unsigned v = 0;
                      for (uint32_t k2 = 4*y; k2 < 4*y+4; k2++)
                        { const char c = k2 < r.rlen ? trk[k2] : 'N';
                          v = (v << 2) | (c == 'R' ? 1u : c == 'H' ? 2u : c == 'D' ? 3u : 0u);
                        }
                      d[y] = (unsigned char)v;
                    }
                }
              o += r.rlen;
              *o++ = '\n';
            }
        });
    }
  };

// ---- input text: memory-mapped plain file or a .gz inflated window by window ------------------------------------
struct TextSource
  { MapFile map; gzFile gz = nullptr; bool is_gz = false;
    size_t pos = 0;                           // plain: next unread byte
    std::vector<char> carry;                  // gz: text of the incomplete last record of the previous window
    bool gz_eof = false;
    bool open(const std::string &path, bool gzip)
    { is_gz = gzip;
      if (!gzip) return map.open(path);
      gz = gzopen(path.c_str(),"r");
      if (gz) gzbuffer(gz,1 << 20);
      return gz != nullptr;
    }
    ~TextSource() { if (gz) gzclose(gz); }
  };

int main(int argc, char **argv)
{ const tp_t t_start = now();
  g_die_fast = true;                       // errors are raised from pipeline threads too: leave without unwinding
  bool verbose = false, seeds = false;
  int nthreads = 4, cov = 0, rlen_opt = 20000;
  std::string tmp_path = "./", fk_root, model_path, source;
  std::vector<std::string> pos;
  for (int i = 1; i < argc; i++)
    { const char *a = argv[i];
      if (a[0] == '-')
        switch (a[1])
        { default:
            for (int k = 1; a[k]; k++)
              { if (a[k] == 'v') verbose = true;
                else if (a[k] == 's') seeds = true;
                else die("%s: -%c is an illegal option\n",PROG,a[k]);
              }
            break;
          case 'T': nthreads = arg_int(a,"Number of threads",true); break;
          case 'c': cov = arg_int(a,"Estimated k-mer coverage",false); break;
          case 'r': rlen_opt = arg_int(a,"Average read length",true); break;
          case 'N': fk_root = a+2; break;
          case 'P': tmp_path = a+2; break;
          case 'M': model_path = a+2; break;
        }
      else
        pos.push_back(a);
    }
  if (pos.empty())
    die("Usage: %s %s\n",PROG,USAGE);
  if (verbose) fprintf(stderr,"Info about inputs:\n");

  std::string path = path_to(pos[0]), root;
  int idx;
  for (idx = 0; idx < 10; idx++)
    { root = root_of(pos[0],EXT[idx]);
      int fd = open((path+"/"+root+EXT[idx]).c_str(),O_RDONLY);
      if (fd >= 0) { close(fd); break; }
    }
  if (idx == 10)
    die("Cannot open %s as a .db|.dam or .f{ast}[aq][.gz] file\n",pos[0].c_str());
  if (pos.size() != 1)
    die(idx <= 1 ? "Only single file is accepted for .db and .dam\n" : "Currently only single file is accepted for FASTX input\n");
  const bool is_db = idx <= 1, is_dam = idx == 1, is_gz = idx >= 6;
  if (seeds && !is_db)
    die("%s: -s writes DAZZ_DB tracks: it needs a .db or .dam input (the reference's seed path has no FASTX mode)\n",PROG);
  if (fk_root.empty()) fk_root = path+"/"+root;
  source = path+"/"+root+EXT[idx];
  const std::string out_path = path+"/"+root+".class";
  if (verbose)
    { fprintf(stderr,"    # of sequence files   = %d\n",1);
      fprintf(stderr,"    First (path,root,ext) = (%s, %s, %s)\n",path.c_str(),root.c_str(),EXT[idx]);
      fprintf(stderr,"    FASTK outputs' root   = %s\n",fk_root.c_str());
      fprintf(stderr,"    Output .class file    = %s/%s.class\n",path.c_str(),root.c_str());
    }
  { std::string tp = tmp_path;                                            // ClassPro.c:466-498 (checked; no temp files are made)
    if (tp[0] != '/')
      { char *cwd = getcwd(NULL,0);
        if (tp[0] == '.')
          { if (tp.size() > 1 && tp[1] == '/') tp = std::string(cwd)+tp.substr(1);
            else if (tp.size() == 1) tp = cwd;
            else die("\n%s: -P option: . not followed by /\n",PROG);
          }
        else tp = std::string(cwd)+"/"+tp;
        free(cwd);
      }
    DIR *d = opendir(tp.c_str());
    if (!d) die("\n%s: -P option: cannot open directory %s\n",PROG,tp.c_str());
    closedir(d);
    if (verbose) fprintf(stderr,"    Temp dir path         = %s\n",tp.c_str());
  }

  Run R;
  R.seeds = seeds; R.verbose = verbose; R.is_db = is_db; R.is_dam = is_dam; R.rlen_opt = rlen_opt; R.model_path = model_path;
  if (!R.P.open(fk_root))
    die("%s: Cannot open %s.prof\n",PROG,fk_root.c_str());
  R.K = R.P.kmer; R.Km1 = R.K-1;
  const int K = R.K;
  if (verbose)
    fprintf(stderr,"    Total # of reads      = %lld\n",(long long)R.P.nreads);

  { int low = 0, high = 0; int64_t il = 0, ih = 0;                         // ClassPro.c:536-554
    std::vector<int64_t> h;
    if (!load_hist(fk_root,&low,&high,&il,&ih,h))
      die("%s: Cannot open %s.hist\n",PROG,fk_root.c_str());
    if (verbose) fprintf(stderr,"Global histogram inspection:\n");
    CPOK(cp_hist_covs(h.data(),low,high,il,ih,cov,&R.hcov,&R.dcov));
    if (verbose)
      fprintf(stderr,cov > 0 ? "    Specified (H,D) cov   = (%d,%d)\n" : "    Estimated (H,D) cov   = (%d,%d)\n",R.hcov,R.dcov);
  }

  // devices: every visible one, or the list in CLASSPRO_DEVICES (an id may repeat: several shards on one GPU)
  std::vector<int> devs;
  if (const char *e = getenv("CLASSPRO_DEVICES"))
    { for (const char *p = e; *p; )
        { char *end; long v = strtol(p,&end,10);
          if (end == p) die("%s: CLASSPRO_DEVICES is not a comma-separated list of device ids\n",PROG);
          devs.push_back((int)v);
          p = (*end == ',') ? end+1 : end;
        }
    }
  else
    { int n = 0;
      HIPOK(hipGetDeviceCount(&n));
      for (int d = 0; d < n; d++) devs.push_back(d);
    }
  if (devs.empty()) die("%s: no HIP device\n",PROG);
  const int ndev = (int)devs.size();
  if (const char *e = getenv("CLASSPRO_BATCH_KBASES")) R.batch_bases = (size_t)atoll(e) << 10;   // diagnostic knob (tests)
  ThreadPool pool(nthreads);
  R.pool = &pool;
  TextSource src;
  DazzDB db;
  MapFile bps;
  if (is_db)                                                               // prepare_db, io.c:123-313
    { db.open(source,is_dam);
      if (R.P.nreads != db.nreads)
        die("Inconsistent # of reads: .prof (%d) != .db (%d)\n",(int)R.P.nreads,db.nreads);
      if (db.maxlen > R.max_rlen) R.max_rlen = db.maxlen;              // a database is sized by its longest read (ClassPro.c:87,110)
      if (!bps.open(path+"/."+root+".bps")) die("%s: Cannot open %s for 'r'\n",PROG,(path+"/."+root+".bps").c_str());
    }
  else if (!src.open(source,is_gz))
    die("%s: Cannot open %s\n",PROG,source.c_str());
  std::vector<std::unique_ptr<Device>> D;
  for (int d = 0; d < ndev; d++)
    { D.emplace_back(new Device());
      D.back()->start(&R,devs[(size_t)d],d);        // HIP start-up, tables and buffers come up while the input is indexed
    }

  // the allocator thread: gives every window's stretch of the output its pages, in order, as soon as it is indexed
  Chan<std::shared_ptr<Window>> to_allocate;
  double t_alloc = 0.;
  std::thread allocator([&]
    { std::shared_ptr<Window> w;
      while (to_allocate.pop(w))
        { const tp_t t0 = now();                      // (a file system without fallocate: the pages come at first touch)
          int64_t lo = w->out_lo;
          { std::unique_lock<std::mutex> lk(R.am);    // what the up-front allocation covers (or will cover) is left to it
            if (!R.pre_done || R.pre_upto > lo)
              { R.acv.wait(lk,[&] { return R.pre_done || R.pre_upto >= w->out_hi; });
                if (R.pre_wait_all) R.acv.wait(lk,[&] { return R.pre_done; });
                if (R.pre_upto > lo) lo = R.pre_upto;
              }
          }
          give_pages(R.out_fd,0,lo,w->out_hi-lo,"the .class output");
          give_pages(R.trk_fd,0,w->trk_lo,w->trk_hi-w->trk_lo,"the .class.data track");
          t_alloc += secs(t0,now());
          { std::lock_guard<std::mutex> lk(R.am); w->allocated = true; }
          R.acv.notify_all();
          w.reset();
        }
    });

  R.out_fd = open(out_path.c_str(),O_RDWR|O_CREAT|O_TRUNC,0644);
  if (R.out_fd < 0) die("Cannot open %s\n",out_path.c_str());
  remember_output(out_path);
  { struct sigaction sa; memset(&sa,0,sizeof(sa)); sa.sa_sigaction = on_sigbus; sa.sa_flags = SA_SIGINFO; sigaction(SIGBUS,&sa,nullptr); }
  if (is_db)
    { // .anno: int nreads, int size = 8, int64 0, then the end offset of every read's data (what merge_anno, io.c:15-68,
      // makes of the per-thread pieces); the offsets only depend on the read lengths, so the file is written up front
      const std::string an = path+"/."+root+".class.anno", dn = path+"/."+root+".class.data";
      FILE *anno = fopen(an.c_str(),"wb");
      R.trk_fd = open(dn.c_str(),O_RDWR|O_CREAT|O_TRUNC,0644);
      if (!anno || R.trk_fd < 0) die("Cannot open .*.class.*\n");
      remember_output(dn);
      const int nr = db.nreads, size = 8; const int64_t zero = 0;
      fwrite(&nr,4,1,anno); fwrite(&size,4,1,anno); fwrite(&zero,8,1,anno);
      int64_t t = 0;
      for (int i = 0; i < db.nreads; i++)
        { t += (db.reads[(size_t)i].rlen+3) >> 2; fwrite(&t,8,1,anno); }
      fclose(anno);
      ClassTrack rep_track;
      rep_track.open(path,root,"rep",db.nreads,0);                         // repeat mask track: header only unless -s (io.c:308-312)
      rep_track.close();
      if (seeds) R.rep.resize((size_t)db.nreads);
    }

  // A plain FASTA's .class is at most twice its size plus a few bytes per record (a FASTQ's: its own size), so the whole
  // output can be given its pages up front by one fallocate thread that nothing disturbs: 16 GB in 0.86 s (19 GB/s),
  // against 2-3.3 s when the same pages are allocated window by window beside formatter threads that fault pages of
  // the same file in (8 Gbases end to end: 2.2 s instead of 3.1 s).  Formatting waits for it; HIP start-up, indexing,
  // staging and the device do not.  CLASSPRO_OUT_ALLOC=window keeps the window-by-window allocation only (also what
  // serves .gz and database inputs, and whatever the estimate did not cover).
  std::thread preallocator;
  { const char *e = getenv("CLASSPRO_OUT_ALLOC");
    if (!is_db && !src.is_gz && src.map.len > 0 && !(e && !strcmp(e,"window")))
      { const int64_t est = (src.map.p[0] == '@' ? 1 : 2)*(int64_t)src.map.len+((int64_t)64 << 20);
        R.pre_done = false; R.pre_wait_all = true;
        preallocator = std::thread([&R,est]
          { const int64_t step = (int64_t)256 << 20;
            for (int64_t a = 0; a < est; a += step)
              { const int64_t n = std::min(step,est-a);
                if (fallocate(R.out_fd,FALLOC_FL_KEEP_SIZE,(off_t)a,(off_t)n) != 0) break;   // (an estimate only: what it could not get
                                                                                             //  is asked for again, and checked, window by window)
                { std::lock_guard<std::mutex> lk(R.am); R.pre_upto = a+n; }
                R.acv.notify_all();
              }
            { std::lock_guard<std::mutex> lk(R.am); R.pre_done = true; }
            R.acv.notify_all();
          });
      }
  }
  const double t_setup = secs(t_start,now());

  // ---- windows of the input -> per-device contiguous read ranges -> batches ---------------------------------------
  FxIndexer fx;
  int64_t next_id = 0, out_pos = 0, trk_pos = 0;
  double t_index = 0.;
  size_t WIN0 = (size_t)ndev << 28, WIN = (size_t)ndev << 30;              // 256 MB per device first, then 1 GB per device
  if (const char *e = getenv("CLASSPRO_WINDOW_KB")) WIN0 = WIN = (size_t)atoll(e) << 10;   // diagnostic knob (tests)
  bool first_window = true;
  for (;;)
    { { std::unique_lock<std::mutex> lk(R.wm);                             // at most three windows alive
        R.wcv.wait(lk,[&] { return R.windows_alive < 3; });
        R.windows_alive++;
      }
      const tp_t ti = now();
      std::shared_ptr<Window> w(new Window(),[&R](Window *p)
        { delete p;
          { std::lock_guard<std::mutex> lk(R.wm); R.windows_alive--; }
          R.wcv.notify_all();
        });
      w->first_id = next_id;
      const size_t want = first_window ? WIN0 : WIN;
      first_window = false;
      bool last = false;
      if (is_db)
        { int64_t bases = 0;
          int i = (int)next_id;
          for (; i < db.nreads && (size_t)bases < want; i++)
            { const DazzRead &rd = db.reads[(size_t)i];
              w->keep.push_back(db.header(i).substr(1));
              FxRec r;
              r.name = w->keep.back().data(); r.name_len = (uint32_t)w->keep.back().size();
              r.cmt = nullptr; r.cmt_len = 0; r.own_cmt = false;
              if ((size_t)rd.boff+(size_t)((rd.rlen+3) >> 2) > bps.len) die("%s: Failed read of .bps file (Load_Read)\n",PROG);
              r.seq = bps.p+rd.boff; r.seq_span = (uint32_t)((rd.rlen+3) >> 2); r.rlen = (uint32_t)rd.rlen;
              w->recs.push_back(r);
              bases += rd.rlen;
            }
          last = i >= db.nreads;
        }
      else
        { const char *text; size_t len; bool eof;
          if (!src.is_gz)
            { text = src.map.p+src.pos;
              len = std::min(want,src.map.len-src.pos);
              eof = src.pos+len >= src.map.len;
            }
          else
            { w->owned.resize(src.carry.size()+want);
              memcpy(w->owned.data(),src.carry.data(),src.carry.size());
              size_t got = src.carry.size();
              while (got < w->owned.size() && !src.gz_eof)
                { int n = gzread(src.gz,w->owned.data()+got,(unsigned)std::min<size_t>(w->owned.size()-got,1u << 30));
                  if (n < 0) die("%s: error reading %s\n",PROG,source.c_str());
                  if (n == 0) { src.gz_eof = true; break; }
                  got += (size_t)n;
                }
              w->owned.resize(got);
              text = w->owned.data(); len = got; eof = src.gz_eof;
              src.carry.clear();
            }
          int status;
          size_t used = fx.index(text,len,eof,pool,w->recs,&status);
          if (status == FX_BADQUAL)
            die("Cannot load %lld-th read\n",(long long)(next_id+(int64_t)w->recs.size()));
          if (used < len)                                                  // the window ends inside a record
            { if (used == 0) die("%s: a record of %s is longer than the input window\n",PROG,source.c_str());
              if (src.is_gz) src.carry.assign(text+used,text+len);
              eof = false;
            }
          if (!src.is_gz) src.pos += used;
          last = eof;
          fx.resolve_comments(w->recs,0,w->keep);
        }
      const size_t nrec = w->recs.size();
      next_id += (int64_t)nrec;
      if (next_id > R.P.nreads)
        die("Inconsistent # of reads: more than the %lld of the .prof\n",(long long)R.P.nreads);
      // record sizes -> final offsets (ClassPro.c:289: header\nseq\n+\nlabels\n)
      w->out_off.resize(nrec+1);
      if (is_db) w->trk_off.resize(nrec+1);
      for (size_t q = 0; q < nrec; q++)
        { const FxRec &r = w->recs[q];
          if (!is_db && (int)r.rlen > CP_MAX_READ_LEN)
            die("rlen (%d) > MAX_READ_LEN for FASTX inputs (%d)\n",(int)r.rlen,CP_MAX_READ_LEN);
          w->out_off[q] = out_pos;
          out_pos += 1+(int64_t)r.name_len+(r.cmt ? 1+(int64_t)r.cmt_len : 0)+1+2*((int64_t)r.rlen+1)+2;
          if (is_db) { w->trk_off[q] = trk_pos; trk_pos += (r.rlen+3) >> 2; }
        }
      w->out_off[nrec] = out_pos;
      if (is_db) w->trk_off[nrec] = trk_pos;
      if (nrec > 0)
        { w->map_out(R.out_fd,w->out_off[0],out_pos,false);
          if (is_db) w->map_out(R.trk_fd,w->trk_off[0],trk_pos,true);
          to_allocate.push(w);
        }
      t_index += secs(ti,now());

      // contiguous ranges per device, balanced by bases; batches inside a range
      std::vector<int64_t> cum(nrec+1,0);
      for (size_t q = 0; q < nrec; q++) cum[q+1] = cum[q]+w->recs[q].rlen;
      size_t a = 0;
      for (int d = 0; d < ndev; d++)
        { size_t z = nrec;
          if (d+1 < ndev)
            { const int64_t target = cum[nrec]*(d+1)/ndev;
              z = (size_t)(std::lower_bound(cum.begin(),cum.end(),target)-cum.begin());
              if (z < a) z = a;
              if (z > nrec) z = nrec;
            }
          size_t b0 = a;
          while (b0 < z)
            { size_t b1 = b0; int64_t bases = 0, codes = 0; size_t reads = 0;
              while (b1 < z)
                { const uint32_t rl = w->recs[b1].rlen;
                  const int64_t cl = R.P.code_len(w->first_id+(int64_t)b1);
                  if (b1 > b0 && ((size_t)(bases+rl) > R.batch_bases || reads+1 > R.batch_reads || (size_t)(codes+cl) > R.batch_bases))
                    break;
                  bases += rl; codes += cl; reads++; b1++;
                }
              D[(size_t)d]->in.push(BatchJob{w,b0,b1});
              b0 = b1;
            }
          a = z;
        }
      if (last) break;
    }
  if (next_id != R.P.nreads && !is_db)
    die("Inconsistent # of reads: .prof (%lld) != input (%lld)\n",(long long)R.P.nreads,(long long)next_id);
  for (auto &d : D) d->in.close();
  to_allocate.close();
  allocator.join();
  if (preallocator.joinable()) preallocator.join();
  for (auto &d : D) { d->feeder.join(); d->completer.join(); }
  if (ftruncate(R.out_fd,(off_t)out_pos) != 0) die("%s: cannot size the output\n",PROG);
  close(R.out_fd);
  if (R.trk_fd >= 0) { if (ftruncate(R.trk_fd,(off_t)trk_pos) != 0) die("%s: cannot size the track\n",PROG); close(R.trk_fd); }
  if (seeds)
    { // .rep.anno: cumulative byte ends of every read's intervals (seed.c:567-570 per thread, merge_anno io.c:15-68 across
      // threads); .rep.data: the (b,e) int pairs.  A read shorter than K has no interval (the reference writes no entry
      // for it at all, ClassPro.c:209-226, which leaves its track one short; here the offset is repeated).
      FILE *ra = fopen((path+"/."+root+".rep.anno").c_str(),"r+b"), *rd = fopen((path+"/."+root+".rep.data").c_str(),"wb");
      if (!ra || !rd) die("Cannot open .*.rep.*\n");
      fseek(ra,16,SEEK_SET);
      int64_t ridx = 0;
      for (const auto &v : R.rep)
        { if (!v.empty()) fwrite(v.data(),4,v.size(),rd);
          ridx += (int64_t)v.size()*4;
          fwrite(&ridx,8,1,ra);
        }
      fclose(ra); fclose(rd);
    }

  if (verbose)
    { const double s = secs(t_start,now());
      fprintf(stderr,"\nResources for phase:  %.3f (s) wall, %.1f Mbases classified (%.1f Mbases/s end to end)\n",
              s,R.total_bases.load()/1e6,R.total_bases.load()/1e6/s);
      fprintf(stderr,"    host: %d device shard(s), %d host threads; %.3f s set-up, %.3f s indexing the input, %.3f s staging, "
                     "%.3f s waiting for the device, %.3f s formatting in place (summed over the pipeline threads); %.3f s in fallocate (its own thread)\n",
              ndev,nthreads,t_setup,t_index,R.t_stage_us.load()/1e6,R.t_wait_us.load()/1e6,R.t_write_us.load()/1e6,t_alloc);
    }
  // the results are on file: leave without tearing down the HIP runtime and 10+ GB of pinned / device buffers
  fflush(stderr);
  _exit(0);
}

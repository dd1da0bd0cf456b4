// classpro_main.cpp -- `ClassPro`-compatible command line on top of the C ABI (libclasspro_amd.so).
//
// Same process contract as the reference binary (src/const.c:14-17, src/ClassPro.c:348-631):
//   ClassPro [-vs] [-T<int(4)>] [-c<int>] [-r<int(20000)>] [-P<tmp_dir(./)>] [-N<fastk_root>]
//            [-M<model_path>] <source>[.db|.dam|.f[ast][aq][.gz]]
// inputs  <fk_root>.hist, <fk_root>.prof, <dir>/.<root>.pidx.N, <dir>/.<root>.prof.N   (FASTK)
// output  <dir>/<root>.class : "@name comment\nseq\n+\nlabels\n" per read              (ClassPro.c:289)
//
// The reference's pthread-per-read-range loop is replaced by read batches: the host parses FASTX
// (kseq semantics) and FASTK profile code strings into pinned staging buffers (decoded on the device by
// cp_decode_profiles), copies
// them with hipMemcpyAsync, calls cp_classify_batch, and writes the records in input order.  While
// the GPU works on one batch the host stages the next one.
//
// Sources: FASTX (kseq semantics) or a Dazzler .db/.dam (dazz_db.h); for a database the labels are also
// written as the DAZZ_DB track .<root>.class.anno/.data (and the header-only .<root>.rep.* mask track).
// Not supported yet: -s (seed selection, seed.c; exits with a message).
// -T is accepted for compatibility; the device does the work, -T only sizes nothing here.
#include <hip/hip_runtime.h>
#include <zlib.h>
#include <dirent.h>
#include <fcntl.h>
#include <unistd.h>
#include <strings.h>
#include <cstdio>
#include <cstdarg>
#include <cstdlib>
#include <cstring>
#include <cstdint>
#include <string>
#include <vector>
#include <chrono>
#include <thread>
#include <mutex>
#include <condition_variable>
#include "../../../include/classpro_amd.h"
#include "host_io.h"
#include "dazz_db.h"

static const char *USAGE = "[-vs] [-T<int(4)>] [-c<int>] [-r<int(20000)>] "
                           "[-P<tmp_dir(./)>] [-N<fastk_root>] [-M<model_path>] "
                           "<source>[.db|.dam|.f[ast][aq][.gz]";                      // const.c:14-17
static const char *EXT[10] = { ".db", ".dam", ".fastq", ".fasta", ".fq", ".fa",
                               ".fastq.gz", ".fasta.gz", ".fq.gz", ".fa.gz" };          // ClassPro.h:326-330

#define HIPOK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) die("%s: %s: %s\n",PROG,#call,hipGetErrorString(e_)); } while (0)
#define CPOK(call)  do { if ((call) < 0) die("%s\n",cp_last_error()); } while (0)

// ---- one batch in flight ----------------------------------------------------------------------------
struct Batch
  { char *h_seq = nullptr, *h_lab = nullptr; uint8_t *h_code = nullptr;
    int64_t *h_soff = nullptr, *h_poff = nullptr, *h_coff = nullptr, *h_boff = nullptr;
    uint8_t *h_pack = nullptr, *d_pack = nullptr;    // database inputs: 2-bit bases as stored in the .bps file
    int64_t *d_boff = nullptr; int64_t packed = 0;
    char *d_seq = nullptr, *d_lab = nullptr; uint8_t *d_code = nullptr; uint16_t *d_prof = nullptr;
    int64_t *d_soff = nullptr, *d_poff = nullptr, *d_coff = nullptr;
    size_t cap_bases = 0, cap_reads = 0, cap_code = 0;
    int n = 0; int64_t bases = 0, kmers = 0, codes = 0;
    std::vector<int64_t> read_id;                // classified read -> 0-based input index (for messages)
    std::vector<std::string> headers;            // every record of the batch, in order (short reads included)
    std::vector<int> slot;                       // record -> index among classified reads, or -1 (short read)
    std::vector<std::string> short_seq;
    hipStream_t st = nullptr;
    cp_workspace *ws = nullptr;
    void alloc(size_t bases_cap, size_t reads_cap, bool with_pack)
    { cap_bases = bases_cap; cap_reads = reads_cap;
      HIPOK(hipHostMalloc((void **)&h_seq,bases_cap)); HIPOK(hipHostMalloc((void **)&h_lab,bases_cap));
      cap_code = bases_cap;                      // FASTK codes run ~0.1-0.3 B/base on HiFi data; a batch closes early if they fill up
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
    void reset() { n = 0; bases = kmers = codes = packed = 0; headers.clear(); slot.clear(); short_seq.clear(); read_id.clear(); }
  };

int main(int argc, char **argv)
{ auto t_start = std::chrono::steady_clock::now();
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
  (void)nthreads;
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
  const bool is_db = idx <= 1, is_dam = idx == 1;
  if (seeds)
    die("%s: -s (seed selection, seed.c) is not supported by this build\n",PROG);
  if (fk_root.empty()) fk_root = path+"/"+root;
  source = path+"/"+root+EXT[idx];
  const std::string out_path = path+"/"+root+".class";
  if (verbose)
    { fprintf(stderr,"    # of sequence files   = %d\n",1);
      fprintf(stderr,"    First (path,root,ext) = (%s, %s, %s)\n",path.c_str(),root.c_str(),EXT[idx]);
      fprintf(stderr,"    FASTK outputs' root   = %s\n",fk_root.c_str());
      fprintf(stderr,"    Output .class file    = %s/%s.class\n",path.c_str(),root.c_str());
    }
  { std::string tp = tmp_path;                                            // ClassPro.c:466-498
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

  Profiles P;
  if (!P.open(fk_root))
    die("%s: Cannot open %s.prof\n",PROG,fk_root.c_str());
  const int K = P.kmer, Km1 = K-1;
  if (verbose)
    fprintf(stderr,"    Total # of reads      = %lld\n",(long long)P.nreads);

  int hcov, dcov;                                                          // ClassPro.c:536-554
  { int low = 0, high = 0; int64_t il = 0, ih = 0;
    std::vector<int64_t> h;
    if (!load_hist(fk_root,&low,&high,&il,&ih,h))
      die("%s: Cannot open %s.hist\n",PROG,fk_root.c_str());
    if (verbose) fprintf(stderr,"Global histogram inspection:\n");
    CPOK(cp_hist_covs(h.data(),low,high,il,ih,cov,&hcov,&dcov));
    if (verbose)
      fprintf(stderr,cov > 0 ? "    Specified (H,D) cov   = (%d,%d)\n" : "    Estimated (H,D) cov   = (%d,%d)\n",hcov,dcov);
  }
  cp_params *params;
  CPOK(cp_params_create_model(K,rlen_opt,hcov,dcov,model_path.empty() ? NULL : model_path.c_str(),&params));
  { int c4[4];
    cp_params_export(params,c4,NULL,NULL,NULL,NULL,NULL,NULL);
    if (verbose)
      { fprintf(stderr,"    Estimated R-threshold = %d\n",c4[CP_REPEAT]);
        if (model_path.empty())
          fprintf(stderr,"Error model not specified. Using the default error model.\n");
        fprintf(stderr,"Classifying %d-mers...\n",K);
      }
  }

  FastxReader fx(is_db ? "/dev/null" : source.c_str());
  if (!fx.f) die("%s: Cannot open %s\n",PROG,source.c_str());
  DazzDB db;
  ClassTrack class_track, rep_track;
  if (is_db)                                                               // prepare_db, io.c:123-313
    { db.open(source,is_dam);
      if (P.nreads != db.nreads)
        die("Inconsistent # of reads: .prof (%d) != .db (%d)\n",(int)P.nreads,db.nreads);
      if (db.maxlen > CP_MAX_READ_LEN)
        die("%s: longest read of the DB (%d) > %d, the longest read this build classifies\n",PROG,db.maxlen,CP_MAX_READ_LEN);
      class_track.open(path,root,"class",db.nreads,8);
      rep_track.open(path,root,"rep",db.nreads,0);                         // repeat mask track: written by -s only
      rep_track.close();
    }
  FILE *out = fopen(out_path.c_str(),"w");
  if (!out) die("Cannot open %s\n",out_path.c_str());
  std::vector<char> obuf(1 << 22);
  setvbuf(out,obuf.data(),_IOFBF,obuf.size());

  // Three batches in flight: the main thread reads batch k+1 from the input while the device classifies
  // batch k and a writer thread prints batch k-1.
  const size_t BATCH_BASES = (size_t)128 << 20, BATCH_READS = 1 << 16;
  constexpr int NB = 3;
  Batch B[NB];
  for (int k = 0; k < NB; k++)
    B[k].alloc(BATCH_BASES+CP_MAX_READ_LEN,BATCH_READS,is_db);

  int64_t id = 0, total_bases = 0;
  bool more = true;
  double t_stage = 0., t_wait = 0., t_write = 0.;
  auto now = [] { return std::chrono::steady_clock::now(); };
  auto secs = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double>(b-a).count(); };
  const double t_setup = secs(t_start,now());
  auto stage = [&](Batch &b)                       // fill one batch from the input; false when nothing was read
    { const auto t0_ = now();
      b.reset();
      b.h_soff[0] = b.h_poff[0] = b.h_coff[0] = 0;
      while (more && (size_t)b.bases < BATCH_BASES && (size_t)b.n < BATCH_READS)
        { if (id >= P.nreads) { more = false; break; }
          if ((size_t)b.codes+2*(size_t)CP_MAX_READ_LEN+2 > b.cap_code) break;
          int rlen;
          if (is_db)                                                       // ClassPro.c:161-180
            { db.load((int)id,fx.seq);
              rlen = (int)fx.seq.size();
              b.headers.push_back(db.header((int)id));
            }
          else
            { rlen = fx.next();
              if (rlen < 0)
                die("Cannot load %lld-th read\n",(long long)id+1);
              if (rlen > CP_MAX_READ_LEN)
                die("rlen (%d) > MAX_READ_LEN for FASTX inputs (%d)\n",rlen,CP_MAX_READ_LEN);
              // header "@name comment": kseq keeps the previous comment when a record has none (ClassPro.c:188)
              b.headers.push_back("@"+fx.name+" "+(fx.have_comment ? fx.comment : std::string("(null)")));
            }
          const uint8_t *code; int64_t clen;
          P.fetch(id,&code,&clen);
          if (rlen <= Km1)                            // ClassPro.c:209-226: printed by the host, not classified
            { b.slot.push_back(-1);
              b.short_seq.push_back(fx.seq);
              id++;
              continue;
            }
          if (clen > 2*(int64_t)CP_MAX_READ_LEN+2)
            die("Read %lld: profile code of %lld bytes is longer than any read of MAX_READ_LEN\n",(long long)id+1,(long long)clen);
          // Fetch_Profile runs on the device (cp_decode_profiles): only the code string crosses PCIe.
          // plen is taken from rlen; the reference's rlen != plen+Km1 check (ClassPro.c:234) is made by
          // the decode kernel and reported in finish().
          int plen = rlen-Km1;
          memcpy(b.h_code+b.codes,code,(size_t)clen);
          memcpy(b.h_seq+b.bases,fx.seq.data(),(size_t)rlen);          // kept on the host for the output record
          if (is_db)
            { if (b.n == 0) b.h_boff[0] = 0;
              memcpy(b.h_pack+b.packed,db.cbuf.data(),(size_t)db.clen);
              b.packed += db.clen;
              b.h_boff[b.n+1] = b.packed;
            }
          b.slot.push_back(b.n);
          b.read_id.push_back(id);
          b.bases += rlen; b.kmers += plen; b.codes += clen; b.n++;
          b.h_soff[b.n] = b.bases; b.h_poff[b.n] = b.kmers; b.h_coff[b.n] = b.codes;
          id++;
        }
      if (id >= P.nreads) more = false;
      t_stage += secs(t0_,now());
      return !b.headers.empty();
    };
  auto submit = [&](Batch &b)
    { if (b.n == 0) return;
      if (!is_db)
        HIPOK(hipMemcpyAsync(b.d_seq,b.h_seq,(size_t)b.bases,hipMemcpyHostToDevice,b.st));
      HIPOK(hipMemcpyAsync(b.d_code,b.h_code,(size_t)b.codes,hipMemcpyHostToDevice,b.st));
      HIPOK(hipMemcpyAsync(b.d_soff,b.h_soff,(size_t)(b.n+1)*8,hipMemcpyHostToDevice,b.st));
      HIPOK(hipMemcpyAsync(b.d_poff,b.h_poff,(size_t)(b.n+1)*8,hipMemcpyHostToDevice,b.st));
      HIPOK(hipMemcpyAsync(b.d_coff,b.h_coff,(size_t)(b.n+1)*8,hipMemcpyHostToDevice,b.st));
      if (is_db)                                                  // 2-bit bases in, characters made on the device
        { HIPOK(hipMemcpyAsync(b.d_pack,b.h_pack,(size_t)b.packed,hipMemcpyHostToDevice,b.st));
          HIPOK(hipMemcpyAsync(b.d_boff,b.h_boff,(size_t)(b.n+1)*8,hipMemcpyHostToDevice,b.st));
          CPOK(cp_unpack_bases(b.d_pack,b.d_boff,b.d_soff,b.n,b.d_seq,b.st));
        }
      CPOK(cp_decode_profiles(b.ws,b.d_code,b.d_coff,b.d_poff,b.n,b.d_prof,b.st));
      CPOK(cp_classify_batch(params,b.ws,b.d_seq,b.d_soff,b.d_prof,b.d_poff,b.n,b.bases,b.kmers,b.d_lab,b.st));
      HIPOK(hipMemcpyAsync(b.h_lab,b.d_lab,(size_t)b.bases,hipMemcpyDeviceToHost,b.st));
    };
  auto wait_device = [&](Batch &b)
    { const auto t0_ = now();
      if (b.n > 0)
        { HIPOK(hipStreamSynchronize(b.st));
          if (cp_workspace_check(b.ws) != CP_OK)
            { // a failed decode: find the read on the host so the message is the reference's (ClassPro.c:234-237)
              std::vector<uint16_t> tmp(CP_MAX_READ_LEN);
              for (int i = 0; i < b.n; i++)
                { int rlen = (int)(b.h_soff[i+1]-b.h_soff[i]);
                  int plen = cp_decode_profile(b.h_code+b.h_coff[i],b.h_coff[i+1]-b.h_coff[i],tmp.data(),CP_MAX_READ_LEN);
                  if (plen >= 0 && rlen != plen+Km1)
                    die("Read %lld: rlen (%d) != plen+Km1 (%d)\n",(long long)b.read_id[i]+1,rlen,plen+Km1);
                }
              die("%s\n",cp_last_error());
            }
        }
      t_wait += secs(t0_,now());
    };
  auto write_out = [&](Batch &b)                   // runs on the writer thread
    { const auto t1_ = now();
      size_t si = 0;
      for (size_t r = 0; r < b.headers.size(); r++)              // ClassPro.c:215,289
        { fputs(b.headers[r].c_str(),out); fputc('\n',out);
          if (b.slot[r] < 0)
            { const std::string &s = b.short_seq[si++];
              fwrite(s.data(),1,s.size(),out); fputs("\n+\n",out);
              for (size_t k = 0; k < s.size(); k++) fputc('N',out);
              fputc('\n',out);
              if (is_db)
                { std::string n(s.size(),'N'); class_track.add(n.data(),(int)n.size()); }
            }
          else
            { int i = b.slot[r];
              size_t o = (size_t)b.h_soff[i], l = (size_t)(b.h_soff[i+1]-b.h_soff[i]);
              fwrite(b.h_seq+o,1,l,out); fputs("\n+\n",out);
              fwrite(b.h_lab+o,1,l,out); fputc('\n',out);
              if (is_db) class_track.add(b.h_lab+o,(int)l);
            }
        }
      total_bases += b.bases;
      t_write += secs(t1_,now());
    };

  // writer thread: prints the batches it is handed, in order
  std::mutex mu;
  std::condition_variable cv;
  int to_write[NB], nq = 0, qhead = 0;             // queue of batch indices
  bool busy[NB] = { false, false, false }, quit = false;
  std::thread writer([&]
    { for (;;)
        { int k;
          { std::unique_lock<std::mutex> lk(mu);
            cv.wait(lk,[&] { return nq > 0 || quit; });
            if (nq == 0) return;
            k = to_write[qhead]; qhead = (qhead+1)%NB; nq--;
          }
          write_out(B[k]);
          { std::lock_guard<std::mutex> lk(mu); busy[k] = false; }
          cv.notify_all();
        }
    });
  auto hand_to_writer = [&](int k)
    { { std::lock_guard<std::mutex> lk(mu); busy[k] = true; to_write[(qhead+nq)%NB] = k; nq++; }
      cv.notify_all();
    };
  auto wait_free = [&](int k)
    { std::unique_lock<std::mutex> lk(mu);
      cv.wait(lk,[&] { return !busy[k]; });
    };

  int cur = 0;
  bool have = stage(B[cur]);
  while (have)
    { submit(B[cur]);
      const int nxt = (cur+1)%NB;
      bool have_next = false;
      if (more)
        { wait_free(nxt);                                    // its previous contents have been printed
          have_next = stage(B[nxt]);                         // the host reads the next batch while the device works
        }
      wait_device(B[cur]);
      hand_to_writer(cur);
      cur = nxt;
      have = have_next;
    }
  { std::lock_guard<std::mutex> lk(mu); quit = true; }
  cv.notify_all();
  writer.join();
  fclose(out);
  class_track.close();

  if (verbose)
    { double s = std::chrono::duration<double>(std::chrono::steady_clock::now()-t_start).count();
      fprintf(stderr,"\nResources for phase:  %.3f (s) wall, %.1f Mbases classified (%.1f Mbases/s end to end)\n",
              s,total_bases/1e6,total_bases/1e6/s);
      fprintf(stderr,"    host: %.3f s set-up, %.3f s reading, %.3f s waiting for the device, %.3f s writing\n",
              t_setup,t_stage,t_wait,t_write);
    }
  cp_params_destroy(params);
  return 0;
}

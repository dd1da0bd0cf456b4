// fx_index_check.cpp -- test harness (CPU): the windowed, chunk-parallel FASTX indexer of the command line
// (classpro_amd/csrc/host/fastx_index.h) against the sequential kseq-semantics reader (host_io.h FastxReader,
// itself pinned by the reference-built tools in tests/test_eval_tools.py) on the same file.
//   fx_index_check <file> <threads> <window_bytes> <carry:0|1>
// carry=1 copies every window into its own buffer together with the unconsumed tail of the previous one (the
// .gz protocol); carry=0 indexes windows of the whole text in place (the mmap protocol).
// Prints "OK <records> <bases>" or the first difference.
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>
#include "../classpro_amd/csrc/host/host_io.h"
#include "../classpro_amd/csrc/host/fastx_index.h"

int main(int argc, char **argv)
{ if (argc < 5) { fprintf(stderr,"usage\n"); return 2; }
  const char *path = argv[1];
  const int nt = atoi(argv[2]);
  const size_t want = (size_t)atoll(argv[3]);
  const bool carry_mode = atoi(argv[4]) != 0;
  // reference parse
  std::vector<std::string> hdr, seq;
  { FastxReader fx(path);
    if (!fx.f) { fprintf(stderr,"cannot open\n"); return 2; }
    for (;;)
      { int n = fx.next();
        if (n < 0) break;
        hdr.push_back("@"+fx.name+" "+(fx.have_comment ? fx.comment : std::string("(null)")));
        seq.push_back(fx.seq);
      }
    if (fx.bad_qual) { printf("BADQUAL %zu\n",hdr.size()); }
  }
  // whole text
  std::string text;
  { FILE *f = fopen(path,"rb");
    char buf[1 << 16]; size_t n;
    while ((n = fread(buf,1,sizeof(buf),f)) > 0) text.append(buf,n);
    fclose(f);
  }
  ThreadPool pool(nt);
  FxIndexer fx;
  if (argc > 5) fx.min_parallel = (size_t)atoll(argv[5]);
  size_t pos = 0, rec = 0; long long bases = 0;
  std::string carry;
  bool done = false;
  while (!done)
    { std::vector<FxRec> recs;
      std::deque<std::string> keep;
      std::string owned;
      const char *t; size_t len; bool eof;
      if (!carry_mode)
        { t = text.data()+pos; len = std::min(want,text.size()-pos); eof = pos+len >= text.size(); }
      else
        { size_t take = std::min(want,text.size()-pos);
          owned = carry+text.substr(pos,take);
          pos += take;
          t = owned.data(); len = owned.size(); eof = pos >= text.size();
          carry.clear();
        }
      int status;
      size_t used = fx.index(t,len,eof,pool,recs,&status);
      if (status == FX_BADQUAL) { printf("BADQUAL %zu\n",rec+recs.size()-1); return 0; }
      if (used < len)
        { if (used == 0) { printf("record longer than window\n"); return 1; }
          if (carry_mode) carry.assign(t+used,t+len);
          eof = false;
        }
      if (!carry_mode) pos += used;
      fx.resolve_comments(recs,0,keep);
      for (const FxRec &r : recs)
        { std::string h = "@"+std::string(r.name,r.name_len);
          if (r.cmt) h += " "+std::string(r.cmt,r.cmt_len);
          std::string s(r.rlen,'\0');
          r.copy_seq(&s[0]);
          if (rec >= hdr.size() || h != hdr[rec] || s != seq[rec])
            { printf("MISMATCH at record %zu: '%s' vs '%s' (len %zu vs %zu)\n",rec,h.c_str(),rec < hdr.size() ? hdr[rec].c_str() : "<none>",
                     s.size(),rec < seq.size() ? seq[rec].size() : (size_t)0);
              return 1;
            }
          rec++; bases += r.rlen;
        }
      done = eof;
    }
  if (rec != hdr.size()) { printf("COUNT %zu vs %zu\n",rec,hdr.size()); return 1; }
  printf("OK %zu %lld\n",rec,bases);
  return 0;
}

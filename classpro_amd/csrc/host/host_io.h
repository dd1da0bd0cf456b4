// host_io.h -- host-side readers shared by the command-line tools (ClassPro, prof2class, class2acc):
// FASTX with kseq.h semantics, the FASTK profile index / histogram files, gene_core-style option
// parsing.  Plain C++; no device code.
#pragma once
#include <zlib.h>
#include <strings.h>
#include <unistd.h>
#include <cstdio>
#include <cstdarg>
#include <cstdlib>
#include <cstring>
#include <cstdint>
#include <string>
#include <vector>

static const char *PROG = "ClassPro";          // each tool sets its own name first thing in main()

static bool g_die_fast = false;               // multi-threaded tools: _exit, so that no thread meets a destroyed object

[[noreturn]] static void die(const char *fmt, ...)
{ va_list ap;
  va_start(ap,fmt);
  vfprintf(stderr,fmt,ap);
  va_end(ap);
  if (g_die_fast) { fflush(stderr); _exit(1); }
  exit(1);
}

// gene_core.c:71-81, 83-109
static std::string path_to(const std::string &name)
{ size_t p = name.rfind('/');
  return p == std::string::npos ? std::string(".") : name.substr(0,p);
}
static std::string root_of(const std::string &name, const char *suffix)
{ size_t p = name.rfind('/');
  std::string f = p == std::string::npos ? name : name.substr(p+1);
  size_t sl = strlen(suffix);
  if (f.size() > sl && strcasecmp(f.c_str()+f.size()-sl,suffix) == 0)
    return f.substr(0,f.size()-sl);
  return f;
}

// ---- integer options, gene_core.h:46-77 ----------------------------------------------------------
static int arg_int(const char *arg, const char *what, bool positive)
{ char *end;
  long v = strtol(arg+2,&end,10);
  if (*end != '\0' || arg[2] == '\0')
    die("%s: -%c '%s' argument is not an integer\n",PROG,arg[1],arg+2);
  if (positive ? v <= 0 : v < 0)
    die("%s: %s must be %s (%ld)\n",PROG,what,positive ? "positive" : "non-negative",v);
  return (int)v;
}

// ---- FASTX reader with kseq.h semantics (name up to the first space, comment = rest of the header
//      line, multi-line sequences, FASTQ qualities skipped) --------------------------------------
struct FastxReader
  { gzFile f;
    std::vector<unsigned char> buf;
    int beg, end;
    bool eof;
    int last;                      // last header char seen ('>' or '@'), 0 = none
    std::string name, comment, seq, qual;
    bool have_comment;
    bool bad_qual = false;         // FASTQ record whose quality string is not as long as its sequence (kseq: -2)

    explicit FastxReader(const char *path) : f(gzopen(path,"r")), buf(1 << 20), beg(0), end(0), eof(false), last(0), have_comment(false)
    { if (f) gzbuffer(f,1 << 20); }
    ~FastxReader() { if (f) gzclose(f); }
    bool fill()
    { if (eof) return false;
      beg = 0;
      end = gzread(f,buf.data(),(unsigned)buf.size());
      if (end <= 0) { eof = true; end = 0; return false; }
      return true;
    }
    int getc()
    { if (beg >= end && !fill()) return -1;
      return buf[beg++];
    }
    // Appends the rest of the current line (without its newline) to *dst, a block at a time; with `graph`
    // only the characters above ' ' (what the per-character loop of the sequence lines keeps).  Returns
    // false when the input ended before a newline.
    bool rest_of_line(std::string *dst, bool graph)
    { for (;;)
        { if (beg >= end && !fill()) return false;
          const unsigned char *p = buf.data()+beg;
          const size_t n = (size_t)(end-beg);
          const unsigned char *nl = (const unsigned char *)memchr(p,'\n',n);
          const size_t len = nl ? (size_t)(nl-p) : n;
          if (dst)
            { bool clean = true;
              if (graph)
                for (size_t k = 0; k < len; k++) clean &= (p[k] > 32);
              if (clean) dst->append((const char *)p,len);
              else
                for (size_t k = 0; k < len; k++)
                  if (p[k] > 32) dst->push_back((char)p[k]);
            }
          beg += (int)(len+(nl ? 1 : 0));
          if (nl) return true;
        }
    }
    // returns sequence length, or -1 at end of file
    int next()
    { int c;
      if (last == 0)
        { while ((c = getc()) != -1 && c != '>' && c != '@') ;
          if (c == -1) return -1;
          last = c;
        }
      name.clear(); seq.clear(); qual.clear();
      bool got_comment = false;
      std::string cm;
      while ((c = getc()) != -1 && c != ' ' && c != '\t' && c != '\n' && c != '\r') name.push_back((char)c);
      if (c == ' ' || c == '\t')
        { got_comment = true;
          rest_of_line(&cm,false);
          if (!cm.empty() && cm.back() == '\r') cm.pop_back();
        }
      else if (c == '\r')
        rest_of_line(nullptr,false);
      if (got_comment) { comment = cm; have_comment = true; }    // kseq leaves the old comment buffer otherwise
      while ((c = getc()) != -1 && c != '>' && c != '+' && c != '@')
        { if (c == '\n') continue;
          if (c > 32) seq.push_back((char)c);
          rest_of_line(&seq,true);
        }
      if (c == '>' || c == '@') last = c;
      else last = 0;
      if (c == '+')                                             // FASTQ: skip the rest of the '+' line, read the qualities
        { if (!rest_of_line(nullptr,false)) { bad_qual = true; return -1; }   // kseq: "no quality string" (-2)
          while (qual.size() < seq.size())                      // whole lines, as kseq does
            { const bool more = rest_of_line(&qual,false);
              if (!qual.empty() && qual.back() == '\r') qual.pop_back();
              if (!more) break;
            }
          last = 0;
          if (qual.size() != seq.size()) { bad_qual = true; return -1; }
        }
      return (int)seq.size();
    }
  };

// ---- FASTK profile index (libfastk.c:1267-1370) ----------------------------------------------------
struct Profiles
  { int kmer = 0, nparts = 0;
    int64_t nreads = 0;
    std::vector<int64_t> index;      // end offset of read i inside its part
    std::vector<int64_t> nbase;      // reads before the end of part p
    std::string prefix;              // <dir>/.<root>.
    int cpart = -1;
    std::vector<uint8_t> data;       // current part, whole file

    bool open(const std::string &fk_root)
    { std::string dir = path_to(fk_root), root = root_of(fk_root,".prof");
      std::string stub = dir+"/"+root+".prof";
      FILE *f = fopen(stub.c_str(),"rb");
      if (!f) return false;
      int smer, nthreads;
      if (fread(&smer,4,1,f) != 1 || fread(&nthreads,4,1,f) != 1) { fclose(f); return false; }
      fclose(f);
      prefix = dir+"/."+root+".";
      index.clear(); nbase.clear(); nreads = 0;
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
        }
      kmer = smer; nparts = nthreads;
      return true;
    }
    // code string of read `id`
    void fetch(int64_t id, const uint8_t **code, int64_t *len)
    { int w = 0;
      while (w < nparts && id >= nbase[w]) w++;
      if (w >= nparts) die("Id %lld is out of range [1,%lld]\n",(long long)id,(long long)nbase[nparts-1]);
      if (w != cpart)
        { std::string nm = prefix+"prof."+std::to_string(w+1);
          FILE *f = fopen(nm.c_str(),"rb");
          if (!f) die("Profile part %s is misssing ?\n",nm.c_str());
          fseek(f,0,SEEK_END); long sz = ftell(f); fseek(f,0,SEEK_SET);
          data.resize((size_t)sz);
          if (sz > 0 && fread(data.data(),1,(size_t)sz,f) != (size_t)sz) die("Cannot read %s\n",nm.c_str());
          fclose(f);
          cpart = w;
        }
      int64_t first = (w == 0) ? 0 : nbase[w-1];
      int64_t off = (id == first) ? 0 : index[(size_t)id-1];
      *code = data.data()+off;
      *len = index[(size_t)id]-off;
    }
  };

static bool load_hist(const std::string &fk_root, int *low, int *high, int64_t *ilow, int64_t *ihigh, std::vector<int64_t> &h)
{ std::string full = path_to(fk_root)+"/"+root_of(fk_root,".hist")+".hist";
  FILE *f = fopen(full.c_str(),"rb");
  if (!f) return false;
  int kmer;
  bool ok = fread(&kmer,4,1,f) == 1 && fread(low,4,1,f) == 1 && fread(high,4,1,f) == 1
            && fread(ilow,8,1,f) == 1 && fread(ihigh,8,1,f) == 1;
  if (ok)
    { h.resize((size_t)(*high-*low)+1);
      ok = fread(h.data(),8,h.size(),f) == h.size();
    }
  fclose(f);
  return ok;
}


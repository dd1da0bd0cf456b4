// prof2class.cpp -- ground-truth .class file from a *relative* FASTK profile (counts of each read
// k-mer in the k-mer table of the true genome): 0 -> E, 1 -> H, 2 -> D, >= 3 -> R.
//
// Same process contract as the reference tool (src/prof2class.c:19, 62-66, 203-253):
//   prof2class <relative_profile>[.prof] <source>[.db|.dam|.f[ast][aq][.gz]]
// writes <dir of profile>/<profile root>.class with one "@name comment\nseq\n+\nlabels\n" record per
// read, K-1 leading 'N's.  Sources: FASTX (kseq semantics) or a Dazzler .db/.dam (dazz_db.h).
#include <fcntl.h>
#include <unistd.h>
#include "host_io.h"
#include "dazz_db.h"
#include "../cp_host_setup.h"

static const char *EXT[10] = { ".db", ".dam", ".fastq", ".fasta", ".fq", ".fa",
                               ".fastq.gz", ".fasta.gz", ".fq.gz", ".fa.gz" };          // prof2class.c:22-24

int main(int argc, char **argv)
{ PROG = "prof2class";
  std::vector<std::string> pos;
  for (int i = 1; i < argc; i++)
    if (argv[i][0] == '-')
      { for (int k = 1; argv[i][k]; k++)                                   // ARG_FLAGS(""): no flags exist
          die("%s: -%c is an illegal option\n",PROG,argv[i][k]);
      }
    else
      pos.push_back(argv[i]);
  if (pos.size() != 2)
    die("Usage: %s <relative_profile>[.prof] <source>[.db|.dam|.f[ast][aq][.gz] \n",PROG);

  std::string out_path = path_to(pos[0])+"/"+root_of(pos[0],".prof")+".class";
  FILE *out = fopen(out_path.c_str(),"w");
  if (!out) die("%s: Cannot open %s for 'w'\n",PROG,out_path.c_str());
  std::string source; int idx;
  for (idx = 0; idx < 10; idx++)
    { source = path_to(pos[1])+"/"+root_of(pos[1],EXT[idx])+EXT[idx];
      int fd = open(source.c_str(),O_RDONLY);
      if (fd >= 0) { close(fd); break; }
    }
  if (idx == 10)
    die("Cannot open %s as a .db|.dam or .f{ast}[aq][.gz] file\n",pos[1].c_str());
  const bool is_db = idx <= 1, is_dam = idx == 1;

  Profiles P;
  if (!P.open(pos[0]))
    die("%s: Cannot open %s as a .prof file\n",PROG,pos[0].c_str());
  FastxReader fx(is_db ? "/dev/null" : source.c_str());
  if (!fx.f) die("%s: Cannot open %s\n",PROG,source.c_str());
  DazzDB db;
  if (is_db)
    { db.open(source,is_dam);
      if (P.nreads != db.nreads)                                           // prof2class.c:111-114
        die("Inconsistent # of reads: .prof (%d) != .db (%d)\n",(int)P.nreads,db.nreads);
    }
  std::vector<char> obuf(1 << 22);
  setvbuf(out,obuf.data(),_IOFBF,obuf.size());

  const int Km1 = P.kmer-1, rlen_max = is_db ? db.maxlen : 60000;          // prof2class.c:154-160
  std::vector<uint16_t> profile(60001);
  std::string asgn, header;
  for (int64_t id = 0; id < P.nreads; id++)
    { int rlen;
      if (is_db)
        { db.load((int)id,fx.seq);
          rlen = (int)fx.seq.size();
          header = db.header((int)id);
        }
      else
        { rlen = fx.next();
          if (rlen < 0) { rlen = 0; fx.seq.clear(); }                      // the reference does not check kseq_read here
          header = "@"+fx.name+" "+(fx.have_comment ? fx.comment : std::string("(null)"));
        }
      if (rlen > rlen_max)
        die("rlen (%d) > rlen_max (%d)\n",rlen,rlen_max);
      const uint8_t *code; int64_t clen;
      P.fetch(id,&code,&clen);
      int plen = cp_host_decode_profile(code,clen,profile.data(),(int)profile.size());
      if (plen > (int)profile.size())                                      // prof2class.c:180-184
        { profile.resize((size_t)plen);
          cp_host_decode_profile(code,clen,profile.data(),plen);
        }
      fputs(header.c_str(),out); fputc('\n',out);
      fwrite(fx.seq.data(),1,fx.seq.size(),out);
      fputs("\n+\n",out);
      if (rlen <= Km1)                                                     // prof2class.c:203-208
        asgn.assign((size_t)rlen,'N');
      else
        { asgn.assign((size_t)Km1,'N');
          for (int i = 0; i < plen; i++)                                   // prof2class.c:210-229
            { uint16_t c = profile[i];
              asgn.push_back(c == 0 ? 'E' : c == 1 ? 'H' : c == 2 ? 'D' : 'R');
            }
        }
      fwrite(asgn.data(),1,asgn.size(),out);
      fputc('\n',out);
    }
  fclose(out);
  return 0;
}

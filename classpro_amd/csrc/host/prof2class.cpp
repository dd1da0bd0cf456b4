// prof2class.cpp -- ground-truth .class file from a *relative* FASTK profile (counts of each read
// k-mer in the k-mer table of the true genome): 0 -> E, 1 -> H, 2 -> D, >= 3 -> R.
//
// Same process contract as the reference tool (src/prof2class.c:19, 62-66, 203-253):
//   prof2class <relative_profile>[.prof] <source>[.db|.dam|.f[ast][aq][.gz]]
// writes <dir of profile>/<profile root>.class with one "@name comment\nseq\n+\nlabels\n" record per
// read, K-1 leading 'N's.  .db/.dam sources are not supported yet.
#include <fcntl.h>
#include <unistd.h>
#include "host_io.h"
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
  if (idx <= 1)
    die("%s: .db/.dam sources are not supported by this build; give the FASTX file\n",PROG);

  Profiles P;
  if (!P.open(pos[0]))
    die("%s: Cannot open %s as a .prof file\n",PROG,pos[0].c_str());
  FastxReader fx(source.c_str());
  if (!fx.f) die("%s: Cannot open %s\n",PROG,source.c_str());
  std::vector<char> obuf(1 << 22);
  setvbuf(out,obuf.data(),_IOFBF,obuf.size());

  const int Km1 = P.kmer-1, rlen_max = 60000;                              // prof2class.c:159
  std::vector<uint16_t> profile(rlen_max+1);
  std::string asgn;
  for (int64_t id = 0; id < P.nreads; id++)
    { int rlen = fx.next();
      if (rlen < 0) { rlen = 0; fx.seq.clear(); }                          // the reference does not check kseq_read here
      if (rlen > rlen_max)
        die("rlen (%d) > rlen_max (%d)\n",rlen,rlen_max);
      const uint8_t *code; int64_t clen;
      P.fetch(id,&code,&clen);
      int plen = cp_host_decode_profile(code,clen,profile.data(),rlen_max+1);
      if (plen > rlen_max+1) plen = rlen_max+1;
      fprintf(out,"@%s %s\n",fx.name.c_str(),fx.have_comment ? fx.comment.c_str() : "(null)");
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

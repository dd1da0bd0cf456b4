// class2acc.cpp -- accuracy of an estimated .class file against a ground-truth .class file.
//
// Same process contract as the reference tool (src/class2acc.c:35-51, 140-326):
//   class2acc [-s] [-e<int>] [-f<int(100)>] [-m<int(0)>] [-n<int(100)>] [-r<int(0)>] [-w<int>]
//             [-p<read_profile[.prof]>] <estimate>.class <truth>.class
// prints (stdout) optional per-read / per-window lines, then the 4x4 confusion matrix (truth rows,
// estimate columns, order E R H D) and the overall / [Normal] / [Repeat] accuracy lines.  The
// classified k-mers of a read are the positions after the leading 'N's of the estimate.
#include "host_io.h"
#include "../cp_host_setup.h"

static const char *USAGE =
"[-s] [-e<int>] [-f<int(100)>] [-m<int(0)>] [-n<int(100)>] [-r<int(0)>] [-w<int>] [-p<read_profile[.prof]>] <estimate>.class <truth>.class\n"
"\n"
"  -e<int> : If specified with a value, classification information is shown for every read that has a misclassification rate larger than this value.\n"
"\n"
"  If `-e` is specified, then the following options become valid (Otherwise ignored):\n"
"\n"
"    -s           : If specified, for each read the ground-truth classification and the estimated classification are shown.\n"
"    -m<int(0)>   : Minimum Repeat-mer rate of a read to be shown.\n"
"    -n<int(100)> : Maximum Repeat-mer rate of a read to be shown.\n"
"\n"
"  -f<int(100)>: Used for real datasets. Ignore every read with an Error-mer rate larger than this value. Use this option in the case where the ground-truth assembly is likely to fail to reconstruct some of the k-mers in the read dataset.\n"
"  -r<int(0)>   : Used for global accuracy calculation. Reads with a Repeat-mer rate larger than this value are regarded as 'Repeat reads'.\n"
"  -w<int> : If specified with a value, instead of each read, accuracy is calculated for each window of the size of this value.\n"
"  -p      : Path to .prof file.\n";

static inline int state_of(char c)                     // class2acc.c:15-30: E (and anything else) 0, R 1, H 2, D 3
{ return c == 'R' ? 1 : c == 'H' ? 2 : c == 'D' ? 3 : 0; }

int main(int argc, char **argv)
{ PROG = "class2acc";
  bool show_lq = false, show_class = false;
  int min_r = 0, max_r = 100, thres_lq = -1, thres_r = 0, window = -1, thres_e = 100;
  const char *prof_root = nullptr;
  std::vector<std::string> pos;
  for (int i = 1; i < argc; i++)
    { const char *a = argv[i];
      if (a[0] != '-') { pos.push_back(a); continue; }
      switch (a[1])
        { default:
            for (int k = 1; a[k]; k++)
              if (a[k] != 's' && a[k] != 'e')
                die("%s: -%c is an illegal option\n",PROG,a[k]);
            break;
          case 's': show_class = true; break;
          case 'm': min_r = arg_int(a,"Min %R-mer per read to show details",false); break;
          case 'n': max_r = arg_int(a,"Max %R-mer per read to show details",false); break;
          case 'e': show_lq = true; thres_lq = arg_int(a,"Min %E-mer per read to show details",false); break;
          case 'f': thres_e = arg_int(a,"Max %E-mer per read to calculate accuracy",false); break;
          case 'r': thres_r = arg_int(a,"Read with %R-mer > this value is regarded as repeat",false); break;
          case 'w': window = arg_int(a,"Size of window = unit of coverage calculation",false); break;
          case 'p': prof_root = a+2; break;
        }
    }
  if (pos.size() != 2)
    die("Usage: %s %s\n",PROG,USAGE);
  FastxReader est(pos[0].c_str());
  if (!est.f) die("%s: Cannot open %s [errno=%d]\n",PROG,pos[0].c_str(),errno);
  FastxReader tru(pos[1].c_str());
  if (!tru.f) die("%s: Cannot open %s [errno=%d]\n",PROG,pos[1].c_str(),errno);
  Profiles P;
  bool have_p = false;
  if (prof_root)
    { if (!P.open(prof_root))
        die("%s: Cannot open %s as a .prof file\n",PROG,prof_root);
      have_p = true;
    }
  const int Km1 = have_p ? P.kmer-1 : -1;
  std::vector<uint16_t> profile(20000);

  int id = 1;
  long long ntot = 0, ncor = 0, nfne = 0;
  long long ntot_normal = 0, ncor_normal = 0, nfne_normal = 0;
  long long ntot_repeat = 0, ncor_repeat = 0, nfne_repeat = 0;
  long long cfm[4][4] = {};
  double cov[2] = { -1, -1 };
  while (est.next() >= 0)
    { if (tru.next() < 0)
        die("# seqs in %s > # seqs in %s\n",pos[0].c_str(),pos[1].c_str());
      if (est.name != tru.name)
        die("Read %d inconsistent names: %s (estimate) vs %s (truth)\n",id,est.name.c_str(),tru.name.c_str());
      const std::string &eq = est.qual, &tq = tru.qual;
      if (!(est.seq.size() == eq.size() && tru.seq.size() == tq.size() && est.seq.size() == tru.seq.size()))
        die("Read %d inconsistent lengths\n",id);
      const int L = (int)tq.size();
      if (have_p)
        { const uint8_t *code; int64_t clen;
          P.fetch((int64_t)id-1,&code,&clen);
          int plen = cp_host_decode_profile(code,clen,profile.data(),(int)profile.size());
          if (plen > (int)profile.size())                                   // class2acc.c:166-170
            { profile.resize((size_t)plen);
              cp_host_decode_profile(code,clen,profile.data(),plen);
            }
          if (plen+Km1 != L)
            die("Read %d inconsist lengths: %ld (estimate) vs %d (profile)\n",id,(long)eq.size(),plen+Km1);
        }
      int i = 0;
      while (i < L && eq[i] == 'N')
        { if (tq[i] != 'N')
            die("Read %d inconsistent # of prefix Ns (= K-1)\n",id);
          i++;
        }
      const int rtot = L-i;
      int rcor = 0, rfne = 0, wcor = 0;
      int rcomp[4] = { 0,0,0,0 }, wcomp[4] = { 0,0,0,0 };
      long long scnts[2] = { 0,0 };
      for (int c = 1; i < L; i++, c++)
        { const char t = tq[i], e = eq[i];
          if (e == t) { rcor++; wcor++; }
          if (t == 'E' && e != 'E') rfne++;
          cfm[state_of(t)][state_of(e)]++;
          switch (t)
            { case 'E': rcomp[0]++; wcomp[0]++; break;
              case 'H': rcomp[1]++; wcomp[1]++; break;
              case 'D': rcomp[2]++; wcomp[2]++; break;
              case 'R': rcomp[3]++; wcomp[3]++; break;
              default:  fprintf(stderr,"Invalid class: %c\n",t); break;
            }
          if (have_p)
            { if (t == 'H')      scnts[0] += profile[i-Km1];
              else if (t == 'D') scnts[1] += profile[i-Km1];
              if (window > 0 && c % window == 0)                            // class2acc.c:224-238
                { cov[0] = (wcomp[1] > 0) ? (double)scnts[0]/wcomp[1] : -1;
                  cov[1] = (wcomp[2] > 0) ? (double)scnts[1]/wcomp[2] : -1;
                  if (cov[0] == -1 || cov[1] == -1 || cov[0] > cov[1]) cov[0] = cov[1] = -1;
                  else cov[1] -= cov[0];
                  fprintf(stdout,"%%error = %4.1lf [H1-cov=%.lf,H2-cov=%.lf]\n",
                          (double)(window-wcor)/window*100,cov[0],cov[1]);
                  scnts[0] = scnts[1] = 0;
                  for (int j = 0; j < 4; j++) wcomp[j] = 0;
                  wcor = 0;
                }
            }
        }
      if ((double)rcomp[0]/rtot*100 > thres_e)                              // class2acc.c:243-246
        { id++; continue; }
      ntot += rtot; ncor += rcor; nfne += rfne;
      if ((double)rcomp[3]/rtot*100 > thres_r)
        { ntot_repeat += rtot; ncor_repeat += rcor; nfne_repeat += rfne; }
      else
        { ntot_normal += rtot; ncor_normal += rcor; nfne_normal += rfne; }
      if (have_p)
        { cov[0] = (rcomp[1] > 0) ? (double)scnts[0]/rcomp[1] : -1;
          cov[1] = (rcomp[2] > 0) ? (double)scnts[1]/rcomp[2] : -1;
          if (cov[0] == -1 || cov[1] == -1 || cov[0] > cov[1]) cov[0] = cov[1] = -1;
          else cov[1] -= cov[0];
        }
      if (show_lq && (double)(rtot-rcor)/rtot*100 >= thres_lq
          && min_r <= (double)rcomp[3]/rtot*100 && (double)rcomp[3]/rtot*100 <= max_r)
        { fprintf(stdout,"Read %6d (%ld bp, %d classes): %%error = %4.1lf [%%E=%4.1lf,%%H=%4.1lf,%%D=%4.1lf,%%R=%4.1lf] [H1-cov=%.lf,H2-cov=%.lf]\n",
                  id,(long)tru.seq.size(),rtot,(double)(rtot-rcor)/rtot*100,
                  (double)rcomp[0]/rtot*100,(double)rcomp[1]/rtot*100,(double)rcomp[2]/rtot*100,(double)rcomp[3]/rtot*100,
                  cov[0],cov[1]);
          if (show_class)
            { fprintf(stdout,"truth: %s\n  est: ",tq.c_str());
              for (int k = 0; k < L; k++)
                fputc(tq[k] != eq[k] ? eq[k] : '-',stdout);
              fputc('\n',stdout);
            }
        }
      id++;
    }
  if (tru.next() >= 0)
    die("# seqs in %s < # seqs in %s\n",pos[0].c_str(),pos[1].c_str());

  static const char stoc[4] = { 'E', 'R', 'H', 'D' };
  fprintf(stdout,"\nConfusion Matrix (Truth\\Est):\n  ");
  for (int i = 0; i < 4; i++) fprintf(stdout,"%15c",stoc[i]);
  fprintf(stdout,"\n");
  for (int i = 0; i < 4; i++)
    { fprintf(stdout,"%c:",stoc[i]);
      for (int j = 0; j < 4; j++) fprintf(stdout,"%15lld",cfm[i][j]);
      fprintf(stdout,"\n");
    }
  fprintf(stdout,"\nAccuracy = %4.2lf %% (= %lld / %lld), FN Error = %4.2lf %%\n",
          (double)ncor/ntot*100,ncor,ntot,(double)nfne/ntot*100);
  fprintf(stdout,"[Normal] Accuracy = %4.2lf %% (= %lld / %lld), FN Error = %4.2lf %%\n",
          (double)ncor_normal/ntot_normal*100,ncor_normal,ntot_normal,(double)nfne_normal/ntot_normal*100);
  fprintf(stdout,"[Repeat] Accuracy = %4.2lf %% (= %lld / %lld), FN Error = %4.2lf %%\n",
          (double)ncor_repeat/ntot_repeat*100,ncor_repeat,ntot_repeat,(double)nfne_repeat/ntot_repeat*100);
  return 0;
}

// Test-only: runs the product's DAZZ track writer (classpro_amd/csrc/host/dazz_db.h, ClassTrack) on the
// label strings of a .class file, so the track format can be checked on a machine without a GPU
// (against DAZZ_DB's own Open_Track through oracle/_ref).  usage: track_harness <dir> <root> <x.class>
#include "../classpro_amd/csrc/host/dazz_db.h"

int main(int argc, char **argv)
{ PROG = "track_harness";
  if (argc != 4) die("usage: track_harness <dir> <root> <x.class>\n");
  FastxReader fx(argv[3]);
  if (!fx.f) die("cannot open %s\n",argv[3]);
  std::vector<std::string> labels;
  while (fx.next() >= 0) labels.push_back(fx.qual);
  ClassTrack t, rep;
  t.open(argv[1],argv[2],"class",(int)labels.size(),8);
  rep.open(argv[1],argv[2],"rep",(int)labels.size(),0);
  rep.close();
  for (auto &l : labels) t.add(l.data(),(int)l.size());
  t.close();
  return 0;
}

// dazz_db.h -- read-only access to a Dazzler database (.db) or map (.dam), and the track writer ClassPro
// uses for its outputs.  Restates what the reference takes from DAZZ_DB's DB.c for this path:
//   Open_DB (DB.c:690-900): stub text file <root>.db|.dam, hidden .<root>.idx (a raw x86-64 DAZZ_DB
//           record followed by one DAZZ_READ record per read, DB.h:287-297, 392-422) and .<root>.bps
//           (2-bit packed bases, 4 per byte, first base in the top bits, DB.c:319-363);
//   Read_DB_Stub (DB.c:478-588) for the per-file read counts and FASTA prologs;
//   Load_Read (DB.c:1232-1298) with ascii = 2 (upper case);
//   the read header ClassPro prints: "@<prolog>/<well>/<first pulse>_<last pulse>" for a .db,
//           the scaffold header line of .<root>.hdr at DAZZ_READ.coff with '>' replaced by '@' for a
//           .dam (ClassPro.c:166-180, prof2class.c:186-200).
// Only whole, untrimmed databases are opened (the reference refuses blocks: io.c:157-160).
#pragma once
#include "host_io.h"

struct DazzRead                        // DB.h:287-297 as laid out by the x86-64 ABI (40 bytes)
  { int32_t origin, rlen, fpulse, pad0;
    int64_t boff, coff;
    int32_t flags, pad1;
  };
static_assert(sizeof(DazzRead) == 40,"DAZZ_READ layout");

struct DazzDB
  { bool is_dam = false;
    int nreads = 0, maxlen = 0;
    int64_t totlen = 0;
    std::vector<DazzRead> reads;
    std::vector<int> findx;            // stub: last read index + 1 of every source file
    std::vector<std::string> prolog;   // stub: FASTA header prolog of every source file
    FILE *bases = nullptr, *hdrs = nullptr;
    int map = 0;
    std::vector<unsigned char> cbuf;

    ~DazzDB() { if (bases) fclose(bases); if (hdrs) fclose(hdrs); }

    // `full` = <dir>/<root>.db or <dir>/<root>.dam (as ClassPro.c:456-463 builds it)
    void open(const std::string &full, bool dam)
    { is_dam = dam;
      std::string dir = path_to(full), root = root_of(full,dam ? ".dam" : ".db");
      FILE *stub = fopen(full.c_str(),"r");
      if (!stub) die("%s: Cannot open %s\n",PROG,full.c_str());
      int nfiles = 0, nblocks = 0, cutoff = 0, all = 0; long long size = 0;
      if (fscanf(stub,"files = %9d\n",&nfiles) != 1)
        die("%s: Stub file (.db) of %s is junk\n",PROG,root.c_str());
      std::vector<char> b1(10100), b2(10100);
      for (int i = 0; i < nfiles; i++)
        { int last;
          if (fscanf(stub,"  %9d %10099s %10099s\n",&last,b1.data(),b2.data()) != 3)
            die("%s: Stub file (.db) of %s is junk\n",PROG,root.c_str());
          findx.push_back(last);
          prolog.push_back(b2.data());
        }
      bool have_blocks = fscanf(stub,"blocks = %9d\n",&nblocks) == 1;
      if (have_blocks && fscanf(stub,"size = %11lld cutoff = %9d all = %1d\n",&size,&cutoff,&all) != 3)
        die("%s: Stub file (.db) of %s is junk\n",PROG,root.c_str());
      if (!have_blocks && !dam)                                      // Read_DB_Stub needs both lines (DB.c:556-560)
        die("%s: Stub file %s is junk\n",PROG,full.c_str());
      fclose(stub);

      std::string idx = dir+"/."+root+".idx";
      FILE *f = fopen(idx.c_str(),"rb");
      if (!f) die("%s: Cannot open %s for 'r'\n",PROG,idx.c_str());
      unsigned char hdr[112];                                        // sizeof(DAZZ_DB) on x86-64
      if (fread(hdr,sizeof(hdr),1,f) != 1)
        die("%s: Index file (.idx) of %s is junk\n",PROG,root.c_str());
      int32_t ureads; memcpy(&ureads,hdr,4);
      memcpy(&maxlen,hdr+32,4);
      memcpy(&totlen,hdr+40,8);
      nreads = ureads;                                               // part 0: ulast-ufirst = ureads (DB.c:808-822)
      reads.resize((size_t)nreads);
      if (nreads > 0 && fread(reads.data(),sizeof(DazzRead),(size_t)nreads,f) != (size_t)nreads)
        die("%s: Index file (.idx) of %s is junk\n",PROG,root.c_str());
      fclose(f);
      std::string bps = dir+"/."+root+".bps";
      bases = fopen(bps.c_str(),"rb");
      if (!bases) die("%s: Cannot open %s for 'r'\n",PROG,bps.c_str());
      if (dam)
        { std::string hn = dir+"/."+root+".hdr";
          hdrs = fopen(hn.c_str(),"r");
          if (!hdrs) die("Cannot open .hdr file %s [errno=%d]\n",hn.c_str(),errno);
        }
      cbuf.resize((size_t)maxlen/4+8);
    }

    // Load_Read(db,i,seq,2): upper-case ASCII.  The packed bytes stay in cbuf (clen of them) for callers that
    // ship them to the device as they are.
    int clen = 0;
    void load(int i, std::string &seq)
    { const DazzRead &r = reads[(size_t)i];
      const int len = r.rlen;
      clen = (len+3) >> 2;
      if ((size_t)clen > cbuf.size()) cbuf.resize((size_t)clen);
      if (fseeko(bases,(off_t)r.boff,SEEK_SET) != 0 || (clen > 0 && fread(cbuf.data(),(size_t)clen,1,bases) != 1))
        die("%s: Failed read of .bps file (Load_Read)\n",PROG);
      static const char letter[4] = { 'A', 'C', 'G', 'T' };
      seq.resize((size_t)len);
      for (int k = 0; k < len; k++)
        seq[(size_t)k] = letter[(cbuf[(size_t)k >> 2] >> (6-2*(k & 3))) & 3];
    }

    std::string header(int i)
    { const DazzRead &r = reads[(size_t)i];
      if (!is_dam)
        { while (map > 0 && i < findx[(size_t)map-1]) map--;
          while (map+1 < (int)findx.size() && i >= findx[(size_t)map]) map++;
          char buf[64];
          snprintf(buf,sizeof(buf),"/%d/%d_%d",r.origin,r.fpulse,r.fpulse+r.rlen);
          return "@"+prolog[(size_t)map]+buf;
        }
      std::string h;
      if (fseeko(hdrs,(off_t)r.coff,SEEK_SET) != 0)
        die("%s: Cannot seek in the .hdr file\n",PROG);
      int c;
      while ((c = fgetc(hdrs)) != EOF && c != '\n') h.push_back((char)c);
      if (h.empty()) h = "@";
      h[0] = '@';
      return h;
    }
  };

// ---- DAZZ_DB data tracks as ClassPro writes them (ClassPro.c:217-223, 290-304; io.c:298-313 for the
//      16-byte .anno header; merge_anno io.c:15-68 makes the per-thread pieces cumulative, which is what
//      writing them in one pass gives directly).  .anno: int nreads, int size = 8, int64 0, then the end
//      offset of every read's data; .data: per read COMPRESSED_LEN(rlen) bytes, 2 bits per base
//      (0 for the first K-1 bases, then E=0 R=1 H=2 D=3: const.c ctos), packed like bases.
struct ClassTrack
  { FILE *anno = nullptr, *data = nullptr;
    int64_t tidx = 0;
    void open(const std::string &dir, const std::string &root, const char *kind, int nreads, int size)
    { std::string an = dir+"/."+root+"."+kind+".anno", dn = dir+"/."+root+"."+kind+".data";
      anno = fopen(an.c_str(),"wb"); data = fopen(dn.c_str(),"wb");
      if (!anno || !data) die("Cannot open .*.%s.*\n",kind);
      const int64_t zero = 0;
      fwrite(&nreads,4,1,anno); fwrite(&size,4,1,anno); fwrite(&zero,8,1,anno);
    }
    // labels: rlen bytes of 'N' (prefix) / E R H D
    void add(const char *labels, int rlen)
    { std::vector<unsigned char> out((size_t)((rlen+3) >> 2),0);
      for (int k = 0; k < rlen; k++)
        { const char c = labels[k];
          const unsigned v = c == 'R' ? 1u : c == 'H' ? 2u : c == 'D' ? 3u : 0u;
          out[(size_t)k >> 2] |= (unsigned char)(v << (6-2*(k & 3)));
        }
      if (!out.empty()) fwrite(out.data(),1,out.size(),data);
      tidx += (int64_t)out.size();
      fwrite(&tidx,8,1,anno);
    }
    void close() { if (anno) fclose(anno); if (data) fclose(data); anno = data = nullptr; }
  };

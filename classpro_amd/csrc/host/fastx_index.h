// fastx_index.h -- record index of a FASTA/FASTQ text held in memory (an mmap'ed file, or a window inflated from
// a .gz), with kseq.h's parsing rules (kseq.h:177-216, the reader ClassPro.c:105-109,181-189 uses):
//   * a record starts at '>' or '@'; the name runs to the first white space; after a blank or tab the rest of the
//     header line is the comment; a record WITHOUT a comment prints the previous record's comment (kseq keeps the
//     old buffer; "(null)" before the first one) -- ClassPro.c:188 formats "@%s %s" from both;
//   * sequence lines run until a line starts with '>', '@' or '+'; characters <= ' ' are dropped;
//   * after '+': the rest of that line is skipped and whole quality lines are read until they are as long as the
//     sequence; a different length is an error (kseq returns -2).
// The reference parses with one kseq stream per thread and re-reads the file from its start up to the thread's
// first read (ClassPro.c:105-109).  Here the text is indexed once, in parallel for FASTA: chunk starts are lines
// that begin with '>' or '@' (always record starts in a file without quality lines), every chunk is parsed by the
// same sequential routine, and the inherited comments are resolved in one pass over the result.  A file whose
// first record has qualities is indexed by one thread (a quality line may begin with '@').
#pragma once
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>
#include <deque>
#include "thread_pool.h"

struct FxRec
  { const char *name, *cmt, *seq;      // cmt: the comment this record prints (own or inherited), not terminated
    uint32_t name_len, cmt_len;
    uint32_t seq_span;                 // bytes of the sequence lines, white space included
    uint32_t rlen;                     // bases
    bool own_cmt;
    bool contiguous() const { return seq_span == rlen || (seq_span == rlen+1 && seq[seq_span-1] == '\n'); }
    void copy_seq(char *dst) const     // the sequence without white space
    { if (contiguous()) { memcpy(dst,seq,rlen); return; }
      const unsigned char *p = (const unsigned char *)seq;
      for (uint32_t k = 0; k < seq_span; k++)
        if (p[k] > 32) *dst++ = (char)p[k];
    }
  };

enum { FX_OK = 0, FX_PARTIAL = 1, FX_BADQUAL = 2 };

// Parses records whose header character lies in [beg,limit); reading may continue up to `end` to finish the last
// one.  `beg` must point at a header character.  Returns the position after the last complete record (== the next
// header character, or `end`).  *status: FX_PARTIAL when the text ended inside a record and `eof` is false (the
// caller re-parses from the returned position with more text), FX_BADQUAL for a quality string of the wrong length.
static size_t fx_parse(const char *buf, size_t beg, size_t limit, size_t end, bool eof, std::vector<FxRec> &out,
                       int *status, bool *saw_qual)
{ *status = FX_OK;
  size_t p = beg;
  while (p < limit)
    { const size_t rec0 = p;
      FxRec r;
      p++;                                               // header character
      r.name = buf+p;
      while (p < end && buf[p] != ' ' && buf[p] != '\t' && buf[p] != '\n' && buf[p] != '\r') p++;
      r.name_len = (uint32_t)(buf+p-r.name);
      r.own_cmt = false; r.cmt = nullptr; r.cmt_len = 0;
      if (p >= end) { if (!eof) { *status = FX_PARTIAL; return rec0; } }
      else if (buf[p] == ' ' || buf[p] == '\t')
        { p++;
          const char *nl = (const char *)memchr(buf+p,'\n',end-p);
          if (!nl && !eof) { *status = FX_PARTIAL; return rec0; }
          size_t e = nl ? (size_t)(nl-buf) : end;
          r.cmt = buf+p; r.own_cmt = true;
          size_t ce = e;
          if (ce > p && buf[ce-1] == '\r') ce--;
          r.cmt_len = (uint32_t)(ce-p);
          p = nl ? e+1 : end;
        }
      else
        { const char *nl = (const char *)memchr(buf+p,'\n',end-p);   // at '\n' or '\r': skip the rest of the line
          if (!nl && !eof) { *status = FX_PARTIAL; return rec0; }
          p = nl ? (size_t)(nl-buf)+1 : end;
        }
      // sequence lines
      r.seq = buf+p;
      uint32_t rlen = 0;
      char term = 0;
      for (;;)
        { if (p >= end) { if (!eof) { *status = FX_PARTIAL; return rec0; } break; }
          const char c = buf[p];
          if (c == '>' || c == '@' || c == '+') { term = c; break; }
          const char *nl = (const char *)memchr(buf+p,'\n',end-p);
          if (!nl && !eof) { *status = FX_PARTIAL; return rec0; }
          const size_t e = nl ? (size_t)(nl-buf) : end;
          const unsigned char *q = (const unsigned char *)buf+p;
          const size_t n = e-p;
          uint32_t good = 0;
          for (size_t k = 0; k < n; k++) good += q[k] > 32;
          rlen += good;
          p = nl ? e+1 : end;
        }
      r.seq_span = (uint32_t)(buf+p-r.seq);
      r.rlen = rlen;
      if (term == '+')
        { *saw_qual = true;
          const char *nl = (const char *)memchr(buf+p,'\n',end-p);   // rest of the '+' line
          if (!nl) { if (!eof) { *status = FX_PARTIAL; return rec0; } *status = FX_BADQUAL; out.push_back(r); return end; }
          p = (size_t)(nl-buf)+1;
          size_t qlen = 0;
          while (qlen < rlen)                                         // whole lines, as kseq does
            { if (p >= end) { if (!eof) { *status = FX_PARTIAL; return rec0; } break; }
              const char *n2 = (const char *)memchr(buf+p,'\n',end-p);
              if (!n2 && !eof) { *status = FX_PARTIAL; return rec0; }
              size_t e = n2 ? (size_t)(n2-buf) : end, ce = e;
              if (ce > p && buf[ce-1] == '\r') ce--;
              qlen += ce-p;
              p = n2 ? e+1 : end;
            }
          if (qlen != rlen) { *status = FX_BADQUAL; out.push_back(r); return p; }
          // kseq then scans forward for the next '>' or '@', wherever it is
          while (p < end && buf[p] != '>' && buf[p] != '@') p++;
        }
      out.push_back(r);
    }
  return p;
}

// first line that starts with '>' or '@' at or after x (or `end`)
static size_t fx_next_header(const char *buf, size_t x, size_t end)
{ if (x == 0 && end > 0 && (buf[0] == '>' || buf[0] == '@')) return 0;
  size_t p = x > 0 ? x-1 : 0;
  for (;;)
    { const char *nl = (const char *)memchr(buf+p,'\n',end-p);
      if (!nl) return end;
      p = (size_t)(nl-buf)+1;
      if (p >= end) return end;
      if (buf[p] == '>' || buf[p] == '@') return p;
    }
}

struct FxIndexer
  { bool fastq = false, decided = false;
    size_t min_parallel = (size_t)1 << 22;   // smaller windows are indexed by the calling thread
    std::string last_comment;           // inherited across windows
    bool have_comment = false;

    // Index the window buf[0,len).  `eof`: no text follows.  Returns the number of bytes consumed (complete
    // records); recs are appended.  *status as fx_parse.
    size_t index(const char *buf, size_t len, bool eof, ThreadPool &pool, std::vector<FxRec> &recs, int *status)
    { *status = FX_OK;
      size_t beg = 0;
      while (beg < len && buf[beg] != '>' && buf[beg] != '@') beg++;      // kseq: skip to the first header character
      if (beg >= len) return eof ? len : 0;
      bool saw_qual = false;
      size_t pos = beg;
      if (!decided)
        { std::vector<FxRec> one;
          size_t p1 = fx_parse(buf,beg,beg+1,len,eof,one,status,&saw_qual);
          if (*status == FX_PARTIAL) return 0;
          decided = true; fastq = saw_qual;
          recs.insert(recs.end(),one.begin(),one.end());
          if (*status != FX_OK) return p1;
          pos = p1;
        }
      if (pos < len)
        { if (fastq || pool.size() == 1 || len-pos < min_parallel)
            pos = fx_parse(buf,pos,len,len,eof,recs,status,&saw_qual);
          else
            { const int nc = pool.size()*4;
              std::vector<size_t> cut(nc+1);
              cut[0] = pos; cut[nc] = len;
              for (int c = 1; c < nc; c++)
                { size_t x = pos+(len-pos)/nc*c;
                  cut[c] = fx_next_header(buf,x,len);
                }
              for (int c = 1; c <= nc; c++) if (cut[c] < cut[c-1]) cut[c] = cut[c-1];
              std::vector<std::vector<FxRec>> part(nc);
              std::vector<int> st(nc,FX_OK);
              std::vector<size_t> endp(nc);
              std::vector<char> sq(nc,0);
              pool.parallel_for(nc,[&](int64_t c)
                { bool q = false;
                  // a chunk that ends at a header line is complete in itself; one that runs to the end of the window is not
                  const bool open_end = cut[c+1] >= len;
                  endp[c] = cut[c] < cut[c+1] ? fx_parse(buf,cut[c],cut[c+1],cut[c+1],open_end ? eof : true,part[c],&st[c],&q)
                                              : cut[c];
                  sq[c] = q;
                });
              for (int c = 0; c < nc; c++)
                { if (sq[c] && !fastq)
                    { // a record with qualities in a file that began as FASTA: chunk starts may be wrong -> one thread
                      std::vector<FxRec> seqr;
                      size_t p2 = fx_parse(buf,pos,len,len,eof,seqr,status,&saw_qual);
                      recs.insert(recs.end(),seqr.begin(),seqr.end());
                      fastq = true;
                      pos = p2;
                      goto resolved;
                    }
                }
              for (int c = 0; c < nc; c++)
                { recs.insert(recs.end(),part[c].begin(),part[c].end());
                  if (st[c] != FX_OK) { *status = st[c]; pos = endp[c]; goto resolved; }
                }
              pos = endp[nc-1];
            }
        }
    resolved:
      return pos;
    }

    // inherited comments (kseq keeps the previous comment buffer): call once per window, in order, before the
    // window's text goes away the last comment is copied
    void resolve_comments(std::vector<FxRec> &recs, size_t from, std::deque<std::string> &keep)
    { static const char nullstr[] = "(null)";
      const char *cur = have_comment ? nullptr : nullstr;
      uint32_t curlen = have_comment ? 0 : 6;
      if (have_comment)
        { keep.push_back(last_comment);                 // owned by the window's keep-alive list
          cur = keep.back().data(); curlen = (uint32_t)keep.back().size();
        }
      for (size_t i = from; i < recs.size(); i++)
        { FxRec &r = recs[i];
          if (r.own_cmt) { cur = r.cmt; curlen = r.cmt_len; have_comment = true; }
          else { r.cmt = cur; r.cmt_len = curlen; }
        }
      if (have_comment) last_comment.assign(cur,curlen);
    }
  };

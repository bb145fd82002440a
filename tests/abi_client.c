/* abi_client.c — an independent client of libmkpileup's record-level C ABI (include/mkpileup.h), written from the header's
 * documentation only: it shares no code with the library.  It plays the part of the Rust caller INTEGRATION.md §3 describes:
 * its own BGZF/BAM reader, its own FASTA reader and motif search, mkp_record views over its own buffers, focus bytes and
 * motif-id combos built as the header documents them, and one mkp_shard_begin / mkp_shard_add_records / mkp_shard_run per
 * reference interval (default 100 kb, the reference's --interval-size).  It writes bedMethyl text with printf.
 * tests/test_gpu_abi_client.py builds it and requires its output to equal mkp_pileup_main's for the same options.
 *
 *   abi_client <in.bam> <ref.fa|-> <out.bed> <all|cpg|cg_cgcg> <default_threshold|none> [interval_size]
 *
 * Test infrastructure (not part of the product).  C99 + zlib. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <zlib.h>

#include "mkpileup.h"

static void die(const char* m) { fprintf(stderr, "abi_client: %s\n", m); exit(1); }

/* ---- BGZF: concatenated gzip members, each with a BC extra field holding its size */
static uint8_t* read_file(const char* path, size_t* n) {
  FILE* f = fopen(path, "rb"); if (!f) die("cannot open input");
  fseek(f, 0, SEEK_END); long sz = ftell(f); fseek(f, 0, SEEK_SET);
  uint8_t* b = (uint8_t*)malloc((size_t)sz + 1); if (!b || fread(b, 1, (size_t)sz, f) != (size_t)sz) die("read error");
  fclose(f); *n = (size_t)sz; return b;
}
static uint8_t* bgzf_inflate_all(const uint8_t* c, size_t n, size_t* out_n) {
  size_t cap = n * 4 + 65536, len = 0, o = 0; uint8_t* out = (uint8_t*)malloc(cap); if (!out) die("oom");
  while (o + 18 <= n) {
    if (c[o] != 31 || c[o + 1] != 139) die("not BGZF");
    size_t xlen = c[o + 10] | (c[o + 11] << 8), x = o + 12, xe = x + xlen; int bsize = -1;
    while (x + 4 <= xe) { size_t sl = c[x + 2] | (c[x + 3] << 8); if (c[x] == 'B' && c[x + 1] == 'C' && sl == 2) bsize = c[x + 4] | (c[x + 5] << 8);
      x += 4 + sl; }
    if (bsize < 0) die("BGZF block without BC field");
    size_t total = (size_t)bsize + 1; uint32_t isize; memcpy(&isize, c + o + total - 4, 4);
    if (len + isize > cap) { cap = (len + isize) * 2; out = (uint8_t*)realloc(out, cap); if (!out) die("oom"); }
    if (isize) {
      z_stream zs; memset(&zs, 0, sizeof(zs)); if (inflateInit2(&zs, -15) != Z_OK) die("zlib");
      zs.next_in = (Bytef*)(c + o + 12 + xlen); zs.avail_in = (uInt)(total - xlen - 20); zs.next_out = out + len; zs.avail_out = isize;
      if (inflate(&zs, Z_FINISH) != Z_STREAM_END) die("corrupt BGZF block");
      inflateEnd(&zs);
    }
    len += isize; o += total;
  }
  *out_n = len; return out;
}

/* ---- FASTA */
typedef struct { char name[256]; char* seq; size_t len; } contig_t;
static contig_t* read_fasta(const char* path, int* n_out) {
  size_t n; uint8_t* b = read_file(path, &n); b[n] = 0;
  contig_t* cs = NULL; int nc = 0; size_t i = 0;
  while (i < n) {
    if (b[i] != '>') { i++; continue; }
    cs = (contig_t*)realloc(cs, sizeof(contig_t) * (size_t)(nc + 1)); contig_t* c = &cs[nc++]; memset(c, 0, sizeof(*c));
    size_t j = i + 1, k = 0; while (j < n && b[j] != '\n' && b[j] != ' ' && b[j] != '\t' && k < 255) c->name[k++] = (char)b[j++];
    while (j < n && b[j] != '\n') j++;
    c->seq = (char*)malloc(n - j + 1); size_t l = 0;
    for (j++; j < n && b[j] != '>'; j++) if (b[j] != '\n' && b[j] != '\r') { char ch = (char)b[j]; if (ch >= 'a' && ch <= 'z') ch = (char)(ch - 32);
      c->seq[l++] = ch; }
    c->len = l; i = j;
  }
  free(b); *n_out = nc; return cs;
}

/* ---- focus bytes as include/mkpileup.h documents them: bits 0-1 strand rule (1 '+', 2 '-', 3 both), bits 2-7 combo index */
typedef struct { mkp_motif_combo c[64]; uint32_t n; } combos_t;
static uint32_t combo_index(combos_t* t, const mkp_motif_combo* c) {
  if (c->n_pos == 0 && c->n_neg == 0) return 0;
  for (uint32_t i = 1; i < t->n; i++) if (memcmp(&t->c[i], c, sizeof(*c)) == 0) return i;
  if (t->n >= 64) die("too many combos");
  t->c[t->n] = *c; return t->n++;
}
typedef struct { const char* pat; int off; } motif_t;
/* forward hits of a palindromic motif inside [s, e) only: a hit straddling the interval end does not count (the reference
 * searches the interval's own slice of the reference) */
static void build_focus(const char* seq, uint32_t s, uint32_t e, const motif_t* motifs, int n_motifs, uint8_t* focus, combos_t* combos) {
  uint32_t w = e - s;
  mkp_motif_combo* per = (mkp_motif_combo*)calloc(w, sizeof(mkp_motif_combo)); uint8_t* rule = (uint8_t*)calloc(w, 1);
  for (int m = 0; m < n_motifs; m++) {
    size_t L = strlen(motifs[m].pat); int fo = motifs[m].off, ro = (int)L - (fo + 1);   /* offset of the modified base on the '-' strand copy */
    for (uint32_t i = s; i + L <= e; i++) {
      if (memcmp(seq + i, motifs[m].pat, L) != 0) continue;
      uint32_t p = i + (uint32_t)fo - s, q = i + (uint32_t)ro - s;
      rule[p] |= 1; per[p].pos_ids[per[p].n_pos++] = (uint8_t)m;
      rule[q] |= 2; per[q].neg_ids[per[q].n_neg++] = (uint8_t)m;
    }
  }
  for (uint32_t i = 0; i < w; i++) {
    if (!rule[i]) { focus[i] = 0; continue; }
    for (int k = 0; k < 4; k++) per[i].pos_delta[k] = -128;   /* no strand combining here */
    focus[i] = (uint8_t)(rule[i] | (combo_index(combos, &per[i]) << 2));
  }
  free(per); free(rule);
}

static double now_s(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec; }

int main(int argc, char** argv) {
  if (argc < 6) die("usage: abi_client <in.bam> <ref.fa|-> <out.bed> <all|cpg|cg_cgcg> <default_threshold|none> [interval_size [intervals_per_batch|file]]");
  const char* mode = argv[4]; uint32_t interval = argc > 6 ? (uint32_t)strtoul(argv[6], NULL, 10) : 100000u;
  /* intervals_per_batch > 1: the batch seam — mkp_batch_run takes that many consecutive intervals (a MultiChromCoordinates) per call */
  /* intervals_per_batch = "file": the file seam — one mkp_process_region per contig, the library reads the BAM itself (device ingest with a .bai) */
  const int file_seam = argc > 7 && strcmp(argv[7], "file") == 0;
  uint32_t per_batch = (argc > 7 && !file_seam) ? (uint32_t)strtoul(argv[7], NULL, 10) : 1u; if (per_batch < 1) per_batch = 1;
  size_t cn, n; uint8_t* comp = read_file(argv[1], &cn); uint8_t* d = bgzf_inflate_all(comp, cn, &n); free(comp);
  if (n < 12 || memcmp(d, "BAM\1", 4) != 0) die("not a BAM");
  size_t o = 4; int32_t l_text; memcpy(&l_text, d + o, 4); o += 4 + (size_t)l_text;
  int32_t n_ref; memcpy(&n_ref, d + o, 4); o += 4;
  char (*names)[256] = calloc((size_t)n_ref, 256); uint32_t* lens = calloc((size_t)n_ref, 4);
  for (int i = 0; i < n_ref; i++) { int32_t ln; memcpy(&ln, d + o, 4); o += 4; memcpy(names[i], d + o, (size_t)(ln < 255 ? ln : 255));
    o += (size_t)ln; memcpy(&lens[i], d + o, 4); o += 4; }
  /* records: views + reference ends */
  size_t cap = 1024, nr = 0; mkp_record* recs = malloc(cap * sizeof(*recs)); int32_t* ends = malloc(cap * 4);
  while (o + 4 <= n) {
    int32_t bs; memcpy(&bs, d + o, 4); const uint8_t* r = d + o + 4; o += 4 + (size_t)bs;
    if (nr == cap) { cap *= 2; recs = realloc(recs, cap * sizeof(*recs)); ends = realloc(ends, cap * 4); }
    mkp_record v; memset(&v, 0, sizeof(v));
    memcpy(&v.tid, r, 4); memcpy(&v.pos, r + 4, 4); v.l_qname = r[8]; uint16_t nc; memcpy(&nc, r + 12, 2); v.n_cigar = nc; memcpy(&v.flag, r + 14, 2);
      memcpy(&v.l_qseq, r + 16, 4);
    v.l_data = bs - 32; v.data = r + 32;
    int64_t rl = 0; for (uint32_t k = 0; k < v.n_cigar; k++) { uint32_t w; memcpy(&w, v.data + v.l_qname + 4 * k, 4); uint32_t op = w & 15;
      if (op == 0 || op == 2 || op == 3 || op == 7 || op == 8) rl += w >> 4;
      }
    ends[nr] = v.pos + (int32_t)(rl > 0 ? rl : 1); recs[nr++] = v;
  }
  int n_contigs = 0; contig_t* fa = NULL; motif_t motifs[2]; int n_motifs = 0;
  if (strcmp(mode, "cpg") == 0) { motifs[0].pat = "CG"; motifs[0].off = 0; n_motifs = 1; }
  else if (strcmp(mode, "cg_cgcg") == 0) { motifs[0].pat = "CG"; motifs[0].off = 0; motifs[1].pat = "CGCG"; motifs[1].off = 2; n_motifs = 2; }
  else if (strcmp(mode, "all") != 0) die("mode must be all, cpg or cg_cgcg");
  if (n_motifs) fa = read_fasta(argv[2], &n_contigs);

  mkp_config cfg; memset(&cfg, 0, sizeof(cfg));
  mkp_ctx* ctx = NULL; if (mkp_ctx_create(&cfg, &ctx) != MKP_OK) die("mkp_ctx_create failed (no gfx950 device?)");
  mkp_caller kc; memset(&kc, 0, sizeof(kc)); kc.max_depth = 8000;
  if (strcmp(argv[5], "none") != 0) kc.default_threshold = strtof(argv[5], NULL);   /* "none": the pass-through caller (threshold 0) */
  if (mkp_set_caller(ctx, &kc) != MKP_OK) die(mkp_last_error(ctx));

  FILE* out = fopen(argv[3], "w"); if (!out) die("cannot open output");
  size_t focus_cap = (size_t)interval * per_batch;
  if (file_seam) for (int t = 0; t < n_ref; t++) if (lens[t] > focus_cap) focus_cap = lens[t];
  uint8_t* focus = n_motifs ? malloc(focus_cap) : NULL; combos_t combos; memset(&combos, 0, sizeof(combos)); combos.n = 1;
  mkp_shard* ivs = malloc(per_batch * sizeof(*ivs)); mkp_rows* brows = malloc(per_batch * sizeof(*brows));
  size_t first = 0; uint64_t total_rows = 0, n_calls = 0; double t_api = 0, t0 = now_s();
  mkp_record* batch = malloc(nr * sizeof(*batch) + sizeof(*batch));
  for (int tid = 0; tid < n_ref; tid++) {
    const char* seq = NULL;
    if (n_motifs) { for (int c = 0; c < n_contigs; c++) if (strcmp(fa[c].name, names[tid]) == 0) {
        if (fa[c].len < lens[tid]) die("FASTA contig shorter than BAM header says");
        seq = fa[c].seq; } if (!seq) die("contig missing from FASTA"); }
    while (first < nr && recs[first].tid >= 0 && recs[first].tid < tid) first++;
    size_t lo = first;
    if (file_seam && lens[tid] > 0) {
      mkp_shard sh; memset(&sh, 0, sizeof(sh)); sh.tid = tid; sh.start = 0; sh.end = lens[tid];
      if (n_motifs) { build_focus(seq, 0, lens[tid], motifs, n_motifs, focus, &combos); sh.focus = focus; sh.combos = combos.c;
        sh.n_combos = combos.n; }
      mkp_rows rows; memset(&rows, 0, sizeof(rows));
      double ta = now_s();
      if (mkp_process_region(ctx, argv[1], &sh, &rows) != MKP_OK) die(mkp_last_error(ctx));
      t_api += now_s() - ta; n_calls++;
      for (uint64_t i = 0; i < rows.n_rows; i++) {
        char name[64]; uint32_t code = rows.code_repr[i];
        int k = (code & 0x80000000u) ? snprintf(name, sizeof(name), "%u", code & 0x7fffffffu) : snprintf(name, sizeof(name), "%c", (char)code);
        if (n_motifs >= 2 && rows.motif_idx[i] >= 0) snprintf(name + k, sizeof(name) - (size_t)k, ",%s,%d", motifs[rows.motif_idx[i]].pat,
            motifs[rows.motif_idx[i]].off);
        float frac = (float)rows.n_mod[i] / (float)rows.n_valid[i];
        fprintf(out, "%s\t%u\t%u\t%s\t%u\t%c\t%u\t%u\t255,0,0\t%u\t%.2f\t%u\t%u\t%u\t%u\t%u\t%u\t%u\n", names[tid], rows.pos[i], rows.pos[i] + 1,
            name, rows.n_valid[i], (char)rows.strand[i],
                rows.pos[i], rows.pos[i] + 1, rows.n_valid[i], (double)(frac * 100.0f), rows.n_mod[i], rows.n_canonical[i], rows.n_other[i],
                    rows.n_delete[i], rows.n_fail[i], rows.n_diff[i], rows.n_nocall[i]);
      }
      total_rows += rows.n_rows;
      continue;
    }
    for (uint32_t s = 0; per_batch > 1 && s < lens[tid];) {
      /* the batch seam: `per_batch` consecutive intervals, their records fetched once, one call */
      uint32_t nb_iv = 0, s0 = s;
      for (; nb_iv < per_batch && s < lens[tid]; nb_iv++) {
        uint32_t e = s + interval < lens[tid] ? s + interval : lens[tid];
        memset(&ivs[nb_iv], 0, sizeof(ivs[nb_iv])); ivs[nb_iv].tid = tid; ivs[nb_iv].start = s; ivs[nb_iv].end = e;
        if (n_motifs) { build_focus(seq, s, e, motifs, n_motifs, focus + (size_t)nb_iv * interval, &combos);
          ivs[nb_iv].focus = focus + (size_t)nb_iv * interval; ivs[nb_iv].combos = combos.c; }
        s = e;
      }
      for (uint32_t k = 0; k < nb_iv; k++) if (n_motifs) ivs[k].n_combos = combos.n;   /* one table for the batch */
      int64_t fs = (int64_t)s0 - 16, fe = (int64_t)s + 16;
      while (lo < nr && recs[lo].tid == tid && (int64_t)ends[lo] <= fs) lo++;
      size_t nb = 0;
      for (size_t i = lo; i < nr && recs[i].tid == tid && (int64_t)recs[i].pos < fe; i++) if ((int64_t)ends[i] > fs) batch[nb++] = recs[i];
      double ta = now_s();
      if (mkp_batch_run(ctx, ivs, nb_iv, batch, (uint32_t)nb, brows) != MKP_OK) die(mkp_last_error(ctx));
      t_api += now_s() - ta; n_calls++;
      for (uint32_t k = 0; k < nb_iv; k++) {
        const mkp_rows rows = brows[k];
        for (uint64_t i = 0; i < rows.n_rows; i++) {
          char name[64]; uint32_t code = rows.code_repr[i];
          int kk = (code & 0x80000000u) ? snprintf(name, sizeof(name), "%u", code & 0x7fffffffu) : snprintf(name, sizeof(name), "%c", (char)code);
          if (n_motifs >= 2 && rows.motif_idx[i] >= 0) snprintf(name + kk, sizeof(name) - (size_t)kk, ",%s,%d", motifs[rows.motif_idx[i]].pat,
              motifs[rows.motif_idx[i]].off);
          float frac = (float)rows.n_mod[i] / (float)rows.n_valid[i];
          fprintf(out, "%s\t%u\t%u\t%s\t%u\t%c\t%u\t%u\t255,0,0\t%u\t%.2f\t%u\t%u\t%u\t%u\t%u\t%u\t%u\n", names[tid], rows.pos[i], rows.pos[i] + 1,
              name, rows.n_valid[i], (char)rows.strand[i],
                  rows.pos[i], rows.pos[i] + 1, rows.n_valid[i], (double)(frac * 100.0f), rows.n_mod[i], rows.n_canonical[i], rows.n_other[i],
                      rows.n_delete[i], rows.n_fail[i], rows.n_diff[i], rows.n_nocall[i]);
        }
        total_rows += rows.n_rows;
      }
    }
    for (uint32_t s = 0; per_batch == 1 && s < lens[tid]; s += interval) {
      uint32_t e = s + interval < lens[tid] ? s + interval : lens[tid];
      int64_t fs = (int64_t)s - 16, fe = (int64_t)e + 16;
      /* records of this contig overlapping [s-16, e+16), file order (what IndexedReader::fetch returns) */
      while (lo < nr && recs[lo].tid == tid && (int64_t)ends[lo] <= fs) lo++;   /* records that end before this interval end before every later one */
      size_t nb = 0;
      for (size_t i = lo; i < nr && recs[i].tid == tid && (int64_t)recs[i].pos < fe; i++) if ((int64_t)ends[i] > fs) batch[nb++] = recs[i];
      mkp_shard sh; memset(&sh, 0, sizeof(sh)); sh.tid = tid; sh.start = s; sh.end = e;
      if (n_motifs) { build_focus(seq, s, e, motifs, n_motifs, focus, &combos); sh.focus = focus; sh.combos = combos.c; sh.n_combos = combos.n; }
      mkp_rows rows; memset(&rows, 0, sizeof(rows));
      double ta = now_s();
      if (mkp_shard_begin(ctx, &sh) != MKP_OK) die(mkp_last_error(ctx));
      if (mkp_shard_add_records(ctx, batch, (uint32_t)nb) != MKP_OK) die(mkp_last_error(ctx));
      if (mkp_shard_run(ctx, &rows) != MKP_OK) die(mkp_last_error(ctx));
      t_api += now_s() - ta; n_calls++;
      for (uint64_t i = 0; i < rows.n_rows; i++) {
        char name[64]; uint32_t code = rows.code_repr[i];
        int k = (code & 0x80000000u) ? snprintf(name, sizeof(name), "%u", code & 0x7fffffffu) : snprintf(name, sizeof(name), "%c", (char)code);
        if (n_motifs >= 2 && rows.motif_idx[i] >= 0) snprintf(name + k, sizeof(name) - (size_t)k, ",%s,%d", motifs[rows.motif_idx[i]].pat,
            motifs[rows.motif_idx[i]].off);
        float frac = (float)rows.n_mod[i] / (float)rows.n_valid[i];
        fprintf(out, "%s\t%u\t%u\t%s\t%u\t%c\t%u\t%u\t255,0,0\t%u\t%.2f\t%u\t%u\t%u\t%u\t%u\t%u\t%u\n", names[tid], rows.pos[i], rows.pos[i] + 1,
            name, rows.n_valid[i], (char)rows.strand[i],
                rows.pos[i], rows.pos[i] + 1, rows.n_valid[i], (double)(frac * 100.0f), rows.n_mod[i], rows.n_canonical[i], rows.n_other[i],
                    rows.n_delete[i], rows.n_fail[i], rows.n_diff[i], rows.n_nocall[i]);
      }
      total_rows += rows.n_rows;
    }
  }
  fclose(out);
  double wall = now_s() - t0;
  fprintf(stderr, "[abi_client] per_batch=%u intervals=%llu rows=%llu api_s=%.3f wall_s=%.3f rows_per_s_api=%.0f\n", per_batch,
      (unsigned long long)n_calls, (unsigned long long)total_rows, t_api, wall, t_api > 0 ? (double)total_rows / t_api : 0.0);
  mkp_ctx_destroy(ctx);
  return 0;
}

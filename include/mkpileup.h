/* mkpileup.h — C ABI of libmkpileup: the MI355X (gfx950) implementation of modkit's `pileup`
 * hot path (MM/ML decode -> threshold caller -> per-position aggregation -> bedMethyl rows).
 *
 * The reference (nanoporetech/modkit v0.4.4, pure Rust) has no FFI on this path.  The seam this
 * library replaces is the Rust function
 *     pub fn process_region_batch(&MultiChromCoordinates, bam_fp, &MultipleThresholdModCaller,
 *                                 &PileupNumericOptions, force_allow, combine_strands, max_depth,
 *                                 Option<&EdgeFilter>, Option<&Vec<SamTag>>)
 *         -> Vec<Result<ModBasePileup, String>>                       (src/pileup/mod.rs:684-716)
 * called from ModBamPileup::run (src/pileup/subcommand.rs:733-753) and consumed by
 * PileupWriter::write (src/writers.rs:159-183).  Each entry point below names the reference item
 * it stands in for.  INTEGRATION.md shows the Rust `extern "C"` binding.
 *
 * Conventions: every function returns MKP_OK (0) or a negative mkp_status and never unwinds;
 * mkp_last_error() gives the message.  A mkp_ctx is NOT thread-safe (one per host worker thread,
 * any number per GPU).  Input buffers stay owned by the caller and may be freed as soon as the
 * call that read them returns.  Output arrays (mkp_rows) are owned by the ctx and stay valid
 * until the next mkp_shard_run/mkp_process_region on that ctx or mkp_ctx_destroy.
 * There is no CPU fallback: inputs the device path does not cover fail with MKP_E_UNSUPPORTED.
 */
#ifndef MKPILEUP_H
#define MKPILEUP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
  MKP_OK = 0,
  MKP_E_INVALID = -1,     /* bad argument / call order */
  MKP_E_IO = -2,          /* file open / parse failure */
  MKP_E_UNSUPPORTED = -3, /* input outside the device path's coverage */
  MKP_E_DEVICE = -4,      /* HIP runtime error, or no gfx950 device */
  MKP_E_NOMEM = -5,
  MKP_E_THRESHOLD = -6    /* threshold estimation failed (e.g. < 2 datapoints, thresholds.rs:18) */
} mkp_status;

typedef struct mkp_ctx mkp_ctx;

typedef struct {
  int32_t device;          /* HIP device ordinal */
  uint32_t tile_positions; /* reference positions per LDS tile; 0 = derive from the LDS budget */
  uint32_t reserved[6];
} mkp_config;

/* mod code encoding = ModCodeRepr (src/mod_base_code.rs:105-109):
 *   Code(char)  -> the code point (e.g. 'm' = 0x6d);  ChEbi(id) -> 0x80000000 | id.
 * Numeric order of this encoding is the reference's row order (derived Ord: Code < ChEbi). */
#define MKP_CODE_CHAR(c) ((uint32_t)(unsigned char)(c))
#define MKP_CODE_CHEBI(id) (0x80000000u | (uint32_t)(id))

typedef struct { uint32_t code_repr; float threshold; } mkp_mod_threshold;

/* Replaces MultipleThresholdModCaller (src/threshold_mod_caller.rs:8-13) + PileupNumericOptions
 * (src/pileup/mod.rs:667-671) + EdgeFilter (src/mod_bam.rs:1634-1639) + the force_allow /
 * combine_strands / max_depth arguments of process_region_batch. Bases are indexed A,C,G,T = 0..3. */
typedef struct {
  float default_threshold;
  float per_base_threshold[4];
  uint8_t has_per_base[4];
  const mkp_mod_threshold* per_mod;
  uint32_t n_per_mod;
  uint32_t numeric_mode;   /* 0 Passthrough, 1 Combine (--combine-mods), 2 Collapse(ReDistribute(collapse_code)) */
  uint32_t collapse_code;  /* code_repr, used when numeric_mode == 2 (--ignore / --preset traditional) */
  uint32_t edge_filter;    /* 0/1 */
  uint32_t edge_start, edge_end, edge_inverted;
  uint32_t force_allow_implicit;
  uint32_t combine_strands;
  uint32_t max_depth;      /* htslib's maxcnt: a shard in which bam_plp_push would drop a record under this cap fails loudly */
} mkp_caller;

/* One alignment record = the fields of htslib's bam1_t the path reads
 * (rust-htslib Record: tid/pos/flags/cigar/seq/aux — src/pileup/mod.rs:783-862, src/mod_bam.rs:1388-1470).
 * `data` is bam1_t::data: qname | cigar u32[n_cigar] | 4-bit seq | qual | aux. */
typedef struct {
  int32_t tid;
  int32_t pos;
  uint16_t flag;
  uint16_t l_qname; /* including the NUL, as bam1_core_t.l_qname */
  uint32_t n_cigar;
  int32_t l_qseq;
  int32_t l_data;
  const uint8_t* data;
} mkp_record;

/* Focus positions of a shard = FocusPositions (src/interval_chunks.rs:32-59) flattened:
 * one byte per reference position of [start,end): bits 0-1 = strand rule (1 = '+' tally only,
 * 2 = '-' tally only, 3 = both, 0 = position not in focus); bits 2-7 = index into `combos`
 * (0 = no motif ids).  focus == NULL means FocusPositions::AllPositions. */
#define MKP_MAX_MOTIF_IDS 4
typedef struct {
  uint8_t n_pos;                          /* motif ids attached to '+' rows (get_positive_strand_motif_ids) */
  uint8_t n_neg;                          /* motif ids attached to '-' rows (get_negative_strand_motif_ids) */
  uint8_t pos_ids[MKP_MAX_MOTIF_IDS];
  uint8_t neg_ids[MKP_MAX_MOTIF_IDS];
  int8_t pos_delta[MKP_MAX_MOTIF_IDS];    /* combine-strands: MotifInfo::negative_strand_position offset; -128 = none */
  uint8_t pad[2];
} mkp_motif_combo;

typedef struct {
  int32_t tid;
  uint32_t start, end;          /* reference window; rows are produced for positions in [start,end) */
  const uint8_t* focus;         /* (end-start) bytes or NULL */
  const mkp_motif_combo* combos;
  uint32_t n_combos;
} mkp_shard;

/* Result rows = PileupFeatureCounts (src/pileup/mod.rs:54-68) as SoA, already in the order
 * ModBasePileup::iter_counts_sorted + the per-position (strand, code) sort would give
 * (src/pileup/mod.rs:440-443, 659-664).  fraction_modified is n_mod as f32 / n_valid as f32. */
typedef struct {
  uint64_t n_rows;
  const uint32_t* pos;
  const uint8_t* strand;      /* '+', '-' or '.' */
  const uint32_t* code_repr;
  const int32_t* motif_idx;   /* -1 = None */
  const uint32_t* n_valid;    /* filtered_coverage */
  const uint32_t* n_mod;
  const uint32_t* n_canonical;
  const uint32_t* n_other;
  const uint32_t* n_delete;
  const uint32_t* n_fail;     /* n_filtered */
  const uint32_t* n_diff;
  const uint32_t* n_nocall;
  /* Log-only counters (the reference prints them at debug level).  Here: kept records of the SHARD whose tags yielded calls /
   * that only contribute coverage.  The reference counts per INTERVAL cache, distinct read names (read_cache.rs:357-365): a read
   * spanning k intervals is counted k times there, once here; the sums over a run differ, the rows do not. */
  uint64_t processed_records;
  uint64_t skipped_records;
  /* --partition-tag (PartitionKey, src/pileup/mod.rs:607-610, 795-815): rows are grouped by key (all rows of key 0, then key 1, ..),
   * genome order inside a key.  partition_key[i] indexes partition_key_names; name 0 is "ungrouped" (PartitionKey::NoKey: no
   * partition tags set, or the read carries none of them); the others are the tag values joined by '_' ("missing" for an absent
   * tag), which is the file stem PartitioningBedMethylWriter uses (src/writers.rs:1029-1080). */
  const uint32_t* partition_key;
  uint32_t n_partition_keys;
  const char* const* partition_key_names;
} mkp_rows;

typedef struct {
  double pack_ms, h2d_ms, kernel_ms, d2h_ms;  /* last shard */
  double decode_kernel_ms, pileup_kernel_ms, gather_kernel_ms;
  uint64_t n_reads, n_events, n_rows, n_tiles, n_positions;
  uint64_t alg_bytes_decode, alg_bytes_pileup; /* SURVEY.md §8(d) algorithmic bytes of the decode and pileup kernels */
  double rows_kernel_ms; /* always 0: rows are emitted by the aggregation kernels straight from LDS (kept for layout compatibility) */
  uint64_t alg_bytes_rows;                     /* 44 B per row */
  /* focus runs on the slot pipeline: feature-stream bytes (one per read and focus position in its span) and
   * SURVEY §8(d)'s literal B_agg = 8 B per coverage / call event + 44 B per row for the aggregation kernel */
  uint64_t stream_bytes, alg_bytes_agg_survey;
  uint32_t slot_pipeline, reserved;
} mkp_stats;

/* ---- lifecycle */
int mkp_ctx_create(const mkp_config* cfg, mkp_ctx** out);
void mkp_ctx_destroy(mkp_ctx* ctx);
const char* mkp_last_error(const mkp_ctx* ctx);
const char* mkp_version(void);
/* ABI revision of THIS header: bumped whenever a struct the caller allocates changes size or layout (mkp_run_report grew in round 5:
 * revision 2; round 6 = 3).  The library writes whole structs of its own revision, so a caller built against another header must not
 * hand it its structs: compare once at start-up —  `mkp_abi_version() == MKP_ABI_VERSION && mkp_run_report_size() == sizeof(mkp_run_report)`
 * (the Rust binding asserts the same on its #[repr(C)] mirror). */
#define MKP_ABI_VERSION 3u
uint32_t mkp_abi_version(void);
size_t mkp_run_report_size(void);
/* Threads of the library's host pool (BGZF inflate, packing, planning, text): the CPUs this process may use — affinity mask and
   cgroup CPU quota — capped at 64; MKP_POOL_THREADS in the environment overrides it.  Rayon's `-t` in the reference
   (subcommand.rs:486-489) is the matching knob. */
unsigned mkp_host_threads(void);

/* ---- caller / options: stands in for the `caller`, `pileup_numeric_options`, `force_allow`,
 *      `combine_strands`, `max_depth`, `edge_filter` arguments of process_region_batch */
int mkp_set_caller(mkp_ctx* ctx, const mkp_caller* caller);

/* ---- partition tags: the `partition_tags: Option<&Vec<SamTag>>` argument of process_region_batch (src/pileup/mod.rs:693).
 * tags = two-character SAM tag names (e.g. "HP", "RG"); n = 0 clears.  Applies to the shards begun afterwards. */
int mkp_set_partition_tags(mkp_ctx* ctx, const char* const* tags, uint32_t n);

/* ---- the hot path on records the host already holds (rust-htslib fetch()+records()):
 * begin a shard, append its records in coordinate order, run.  Replaces the body of
 * process_region (src/pileup/mod.rs:718-1020) for every interval inside the shard. */
int mkp_shard_begin(mkp_ctx* ctx, const mkp_shard* shard);
int mkp_shard_add_records(mkp_ctx* ctx, const mkp_record* recs, uint32_t n);
/* Optional, between begin and run: the ascending start positions of the reference's intervals inside the shard (the last one ends with
 * the window).  The reference keeps one read cache per interval, keyed by read NAME (src/read_cache.rs:28-35): two kept records with one
 * name interfere only when they overlap a common interval — there the record asked about first owns the name and the later ones are
 * answered from its calls (get_mod_call, src/read_cache.rs:232-297), which the library reproduces (round 6; refused before); mates / split
 * alignments lying in different intervals are independent reads.  Without this call the whole shard counts as one interval.  Still
 * MKP_E_UNSUPPORTED: pileup-hemi with such records, and a record whose owners in different intervals disagree in status or codes. */
int mkp_shard_set_intervals(mkp_ctx* ctx, const uint32_t* interval_starts, uint32_t n_intervals);
int mkp_shard_run(mkp_ctx* ctx, mkp_rows* out);

/* ---- the same seam one level up: process_region_batch(&MultiChromCoordinates, ...) (src/pileup/mod.rs:684-716, called per batch from
 * ModBamPileup::run, src/pileup/subcommand.rs:733-753) in ONE call.  intervals = the batch's ChromCoordinates (each a mkp_shard: window +
 * focus bytes of that interval), in the order the feeder produced them; recs = the records the caller fetched for the batch — every
 * record overlapping any of the intervals once, coordinate order per contig.  Intervals that follow each other on one contig
 * (start == previous end, same kind of focus, same combo table) are merged into ONE resident shard — one pack, one plan, one launch
 * sequence, one read-back instead of one per interval — and cut apart again at their ends: out[k] = the rows of intervals[k], arrays
 * owned by the ctx until its next run.  processed / skipped_records are reported on the first interval of each merged run.  With
 * partition tags set every interval runs alone (its rows come grouped by key).  Per-interval calls cost ~1.7 ms of launch and planning
 * latency each (bench.py: tiers.seam_per_interval); a batch pays it once per contig run. */
int mkp_batch_run(mkp_ctx* ctx, const mkp_shard* intervals, uint32_t n_intervals, const mkp_record* recs, uint32_t n_recs, mkp_rows* out);

/* Re-run the device pipeline on the shard that is already resident in HBM (no pack, no H2D):
 * what bench.py times.  `iters` launches; rows of the last one are returned. */
int mkp_shard_rerun(mkp_ctx* ctx, uint32_t iters, mkp_rows* out);
int mkp_get_stats(const mkp_ctx* ctx, mkp_stats* out);

/* ---- same, reading the BAM itself: direct stand-in for process_region_batch(bam_fp, ...)
 * (src/pileup/mod.rs:684-716; the IndexedReader fetch + pileup of :732-759).  With a .bai next to the file the window's
 * compressed BGZF blocks are uploaded and inflated, cut into records, tokenised and packed ON THE DEVICE: nothing
 * but the index and the block headers is read on the host, and this is the fast seam — the record-level calls above pay the
 * host packer and the upload of packed records.  Without an index (or with partition tags, or MKP_HOST_INGEST=1) the library's
 * host reader + packer feed the same kernels.  One call = one shard: keep the window to ~1 GiB of BAM. */
int mkp_process_region(mkp_ctx* ctx, const char* bam_path, const mkp_shard* shard, mkp_rows* out);

/* ---- whole subcommand: `modkit pileup` (ModBamPileup::run, src/pileup/subcommand.rs:382-816).
 * argv carries the reference's own flags (in.bam out.bed --cpg --ref ... ), see DESIGN.md for the
 * covered set; plus --device N, --gpus-rank R --gpus-world W for interval sharding. */
int mkp_pileup_main(int argc, const char* const* argv, char* errbuf, size_t errbuf_len);

/* The same subcommand on a context the caller owns (so the device named by the ctx is used and --device is ignored):
 * afterwards the last shard is still resident in HBM — mkp_shard_rerun re-launches the kernels on it — and `report`
 * (may be NULL) holds the wall time of every stage of ModBamPileup::run as this library executes it. */
typedef struct {
  double load_ms;       /* BAM open + BGZF inflate + record index (reference: htslib IndexedReader per interval) */
  double threshold_ms;  /* threshold estimation (sampling schedule + device decode of the sampled reads + percentile) */
  double focus_ms;      /* interval grid + motif / BED focus positions (interval_chunks.rs) */
  double pack_ms, h2d_ms, kernel_ms, d2h_ms;   /* summed over shards */
  double write_ms;      /* bedMethyl text + file write the caller waited for */
  double total_ms;
  uint64_t n_rows, n_positions, n_shards, processed_records, skipped_records;
  float threshold[4]; uint8_t has_threshold[4];   /* per-base pass thresholds used (A,C,G,T) */
  /* round 5 — where the wall went that the stages above do not own, so that they add up; and the device ingest's own figures */
  double grid_wait_ms;          /* the threshold estimate waiting for the reference FASTA and the interval grid (not part of threshold_ms) */
  double callback_ms;           /* mkp_pileup_run_cb: inside the caller's threshold callback (all-reduce / broadcast; not part of threshold_ms) */
  double ingest_kernel_ms;      /* device ingest, summed over the shards: the inflate launch + record chains (HIP events) */
  double ingest_upload_ms, ingest_table_ms, ingest_pack_ms;
    /* ... the uploads, the block tables, parse + scans + pack incl. their syncs (host clocks) */
  uint64_t ingest_comp_bytes, ingest_raw_bytes, ingest_blocks, ingest_records;   /* compressed bytes uploaded, bytes they inflated to */
} mkp_run_report;
int mkp_pileup_run(mkp_ctx* ctx, int argc, const char* const* argv, mkp_run_report* report);

/* The same run with the pass thresholds decided by the CALLER at the point where `modkit pileup` estimates them (subcommand.rs:615-638) —
 * the multi-GPU form: every rank calls this with --gpus-rank / --gpus-world and no threshold flags.  The library ingests this rank's shards
 * ahead from the first moment, then calls `fn` once:
 *   have_sample = 1  (`-f 1.0`, thresholds.rs:121-159): the context's histograms hold the sample of THIS rank's shards, taken from HBM;
 *                    the callback sums them over the ranks (mkp_histogram_allreduce, or mkp_histogram_get + its own collective) and
 *                    evaluates the percentile (mkp_histogram_locate / _resolve / mkp_percentile_from_histogram);
 *   have_sample = 0  (the count-based default, `-n`, `-f x < 1`): nothing was sampled — the schedule carries quotas from interval to
 *                    interval and is one rank's job (mkp_estimate_thresholds on a context of its own) — the callback hands over the
 *                    values it received from that rank.
 * The callback fills thresholds[b] / has[b] for b = A, C, G, T and returns MKP_OK; the shards are still resident when it returns and the
 * pileup pass runs on them.  With fn == NULL this is mkp_pileup_run. */
typedef int (*mkp_threshold_fn)(void* user, mkp_ctx* ctx, int have_sample, float thresholds[4], uint8_t has[4]);
int mkp_pileup_run_cb(mkp_ctx* ctx, int argc, const char* const* argv, mkp_threshold_fn fn, void* user, mkp_run_report* report);

/* ---- threshold estimation: get_threshold_from_options (src/command_utils.rs:74-134) ->
 * calc_threshold_from_bam (src/thresholds.rs:121-159).  Which reads are sampled follows the reference's
 * schedule (reads_sampler/, sampling_schedule.rs); their call probabilities are decoded on the GPU.
 * argv = the sampling flags of `modkit pileup` (-n -f -p -t --sampling-interval-size --region --sample-region
 * --include-bed --include-unmapped --edge-filter --ignore --preset). thr/has are indexed A,C,G,T. */
int mkp_estimate_thresholds(mkp_ctx* ctx, const char* bam_path, int argc, const char* const* argv, float thr[4], uint8_t has[4]);

/* ---- threshold arithmetic: percentile_linear_interp (src/thresholds.rs:17-38) on a sorted array */
int mkp_percentile(const float* sorted, uint64_t n, float q, float* out);

/* ---- the same estimate split for multi-GPU runs: calc_threshold_from_bam (src/thresholds.rs:121-159) sorts the
 * sampled argmax probabilities of each canonical base and interpolates between two order statistics.  Here the sample stays in
 * HBM and is summarised by two-level histograms of the values' f32 bit patterns (positive floats order like unsigned integers):
 * level 0 counts the top 16 bits, level 1 the low 16 bits of the values whose top 16 bits equal `prefix`.  Histograms add, so
 * W ranks that each sampled their own reference windows sum them (ncclAllReduce(sum, u64) over xGMI — the path's one
 * collective) and every rank finds the exact order statistics of the union.  Sequence per rank:
 *   mkp_histogram_begin; mkp_histogram_add_bam(..., "--gpus-rank R --gpus-world W -f 1.0 ...");
 *   per base: get(level 0) -> all-reduce -> mkp_histogram_locate -> get(level 1, bins) -> all-reduce -> mkp_histogram_resolve
 *             -> mkp_percentile_from_histogram.                 (mkp_estimate_thresholds does all of it for one rank.)
 * Histogram arrays are uint64_t[65536]. */
int mkp_histogram_begin(mkp_ctx* ctx);
int mkp_histogram_add_bam(mkp_ctx* ctx, const char* bam_path, int argc, const char* const* argv);
int mkp_histogram_get(mkp_ctx* ctx, uint32_t base /*A,C,G,T = 0..3*/, uint32_t level, uint32_t prefix, uint64_t* out);
/* mkp_histogram_get + the all-reduce in one call, on the device: the histogram is widened to u64 in HBM and summed over the ranks of
 * `nccl_comm` (an ncclComm_t the caller created with ncclCommInitRank, one rank per GPU) with ncclAllReduce(ncclUint64, ncclSum) over
 * xGMI on the context's stream; `out` receives the sum.  Every rank must make the same calls in the same order.  librccl.so is loaded
 * at run time.  (modkit_amd.distributed uses it when the job runs on the nccl backend; torch.distributed's all_reduce is the test
 * double on gloo.) */
int mkp_histogram_allreduce(mkp_ctx* ctx, void* nccl_comm, uint32_t base, uint32_t level, uint32_t prefix, uint64_t* out);
int mkp_histogram_from_values(const float* vals, uint64_t n, uint32_t level, uint32_t prefix, uint64_t* out);  /* host values, no device */
int mkp_histogram_locate(const uint64_t* hist0, float q, uint32_t bins[2], uint64_t ranks_in_bin[2], uint64_t* n);
int mkp_histogram_resolve(uint32_t prefix, const uint64_t* hist1, uint64_t rank_in_bin, float* value);
int mkp_percentile_from_histogram(uint64_t n, float q, float y_floor, float y_ceil, float* out);

/* ---- pileup-hemi: duplex (hemi-methylation) pattern counts.  Stands in for process_region_duplex_batch
 * (src/pileup/duplex.rs:209-339) over every interval inside a shard, with DuplexReadCache::get_duplex_mod_call
 * (src/read_cache.rs:422-462) and DuplexFeatureVector::decode (duplex.rs:124-205) on the device.
 * Rows = DuplexPatternCounts (duplex.rs:32-56) + the position's n_delete, SoA, in the order the reference's writer emits them
 * (position, primary base by letter, pattern — src/writers.rs:185-258).  A pattern element is MKP_HEMI_CANONICAL for '-'
 * (a canonical call) or the mod code (code_repr encoding above); DuplexModCodeRepr's order is the numeric order of that. */
#define MKP_HEMI_CANONICAL 0u
typedef struct {
  uint64_t n_rows;
  const uint32_t* pos;
  const uint8_t* primary_base;      /* 'A','C','G','T': the record's SEQ base at the position, reference orientation */
  const uint32_t* pattern_pos;      /* call on the positive strand at `pos` */
  const uint32_t* pattern_neg;      /* call on the negative strand at the partner position */
  const uint32_t* n_valid;          /* valid_coverage = count + n_other_pattern */
  const uint32_t* count;
  const uint32_t* n_canonical;
  const uint32_t* n_other_pattern;
  const uint32_t* n_delete;
  const uint32_t* n_fail;
  const uint32_t* n_diff;
  const uint32_t* n_nocall;
  uint64_t processed_records, skipped_records;
} mkp_hemi_rows;
/* Run the shard begun with mkp_shard_begin / mkp_shard_add_records as pileup-hemi.  The shard's focus bytes must be those of
 * FocusPositions::MotifCombineStrands for ONE palindromic motif (rule bit 0 + a combo with a positive motif id mark the positions
 * that get rows).  partner_offset = MotifInfo::negative_strand_position(p) - p (src/motif_bed.rs:124-140; 1 for CG,0).
 * interval_starts = ascending start positions of the reference's intervals inside [start,end) (ReferenceIntervalsFeeder,
 * src/interval_chunks.rs:497-652): the reference keeps one read cache per interval, which shows in the output only through
 * records whose tags fail to parse (one NoCall per such record and interval); NULL / 0 = the shard is one interval.
 * mkp_shard_rerun re-launches the hemi kernels afterwards (pass out = NULL there). */
int mkp_hemi_shard_run(mkp_ctx* ctx, int32_t partner_offset, const uint32_t* interval_starts, uint32_t n_intervals, mkp_hemi_rows* out);
/* `modkit pileup-hemi <in.bam> -o <out.bed> [flags]` (DuplexModBamPileup::run, src/pileup/subcommand.rs:1122-1514):
 * argv = the arguments after the subcommand name; without -o the rows go to stdout. */
int mkp_pileup_hemi_main(int argc, const char* const* argv, char* errbuf, size_t errbuf_len);
int mkp_pileup_hemi_run(mkp_ctx* ctx, int argc, const char* const* argv, mkp_run_report* report);

/* ---- `modkit sample-probs` (SampleModBaseProbs, src/commands.rs:549-887), the percentiles table: the reads the reference's schedule
 * samples are decoded on the device (the sampling kernels of the threshold estimate); per canonical base the requested percentiles of
 * their argmax probabilities come out of the HBM-resident sample exactly (Percentiles::new -> percentile_linear_interp,
 * src/thresholds.rs:17-38).  argv = the subcommand's sampling flags (-n -f --no-sampling --region -i --include-bed --only-mapped
 * --ignore --edge-filter --invert-edge-filter -t); values[b * n_percentiles + k] for bases A,C,G,T; has[b] = 0 when the sample holds
 * no call on base b; n_values[b] = sampled calls on base b.  A base with fewer than two values fails with MKP_E_THRESHOLD, as the
 * reference does.  (The histogram / plot outputs of the subcommand are outside this path.)
 * A BAM without an index file takes the reference's serial branch (src/reads_sampler/mod.rs:129-158), as mkp_summary and the estimate of
 * mkp_extract_calls_main do: no schedule, the file in file order under RecordSampler — the first -n records that yield values, or with
 * -f < 1 one `gen_bool` per record from rand's StdRng, which needs --seed (src/reads_sampler/record_sampler.rs:29-38, 80-86; without a
 * seed the reference draws from entropy: MKP_E_UNSUPPORTED); --region is then an error, as in the reference.  With an index, -f < 1 only
 * draws for the records without coordinates (--seed again). */
int mkp_sample_probs(mkp_ctx* ctx, const char* bam_path, int argc, const char* const* argv, const float* percentiles, uint32_t n_percentiles,
                     float* values, uint8_t has[4], uint64_t n_values[4]);

/* ---- `modkit summary` (ModSummarize, src/commands.rs:888-1189 -> summarize_modbam / sampled_reads_to_summary, src/summarize.rs:59-262):
 * ModSummary as counts.  The sampled reads are decoded twice by the sampling kernels: once to estimate the pass thresholds (skipped with
 * --no-filtering / --filter-threshold), once to count every sampled call under its thresholded call — or, when that is Filtered, under
 * its argmax call.  argv = the subcommand's flags (-n -f --no-sampling --region -i --include-bed --only-mapped --ignore --edge-filter
 * --invert-edge-filter -p --no-filtering --filter-threshold --mod-thresholds -t).  Rows: for every canonical base with sampled calls
 * the canonical state (code_repr = MKP_HEMI_CANONICAL) then every observed mod code in code order; arrays owned by the ctx until its
 * next summary call.  (The table / TSV writers iterate std HashMaps and print through f32 Display: they stay in Rust.) */
typedef struct {
  uint64_t total_reads_used;
  uint64_t reads_with_mod_calls[4];     /* per canonical base A,C,G,T */
  float threshold[4]; uint8_t has_threshold[4];
  uint32_t n_rows;
  const uint8_t* base;                  /* 0..3 */
  const uint32_t* code_repr;
  const uint64_t* pass_count;           /* mod_call_counts */
  const uint64_t* fail_count;           /* filtered_mod_call_counts */
} mkp_summary_out;
int mkp_summary(mkp_ctx* ctx, const char* bam_path, int argc, const char* const* argv, mkp_summary_out* out);

/* ---- `modkit extract calls <in.bam> <out.tsv> [flags]` (EntryExtractCalls::run, src/extract/subcommand.rs:452-761): the per-read call table
 * (PositionModCalls::header / to_row, src/extract/writer.rs:12-132) — 21 tab-separated columns per call: read id, forward and reference
 * position, strands, soft clips, read length, call_prob / call_code (BaseModProbs::argmax_base_mod_call), base quality, reference and
 * query k-mers, canonical / modified primary base, fail (MultipleThresholdModCaller::call == Filtered), inferred, within_alignment, flag.
 * The calls (MM / ML decode, edge filter, --ignore collapse, argmax and thresholded call) are computed on the device; ids, positions,
 * qualities, k-mers and the text on the host.  argv = in.bam out.tsv + --ref fa, --allow-non-primary, --mapped-only, --pass-only,
 * --no-headers, --kmer-size k, --no-filtering | --filter-threshold .. | the sampling flags of the threshold estimate (-n -f -p -t
 * --sampling-interval-size), --mod-thresholds, --ignore, --edge-filter, --invert-edge-filter, --device N.  Records go out in FILE order
 * (the reference's serial path, which its golden tests pin; its indexed path emits interval batches in Rayon completion order).
 * Round 6: --include-bed (ReferencePositionFilter::keep, src/extract/util.rs:44-69: rows are asked of the BED with the reference strand
 * of the mod; rows without a reference position go), --region (src/extract/util.rs:126-160: steers the threshold estimate; with a BAI next
 * to the BAM and without --ignore-index the table holds the records overlapping the region — what the reference's interval fetches
 * return — otherwise every record of the file, as its serial scan does), --num-reads N with --ignore-index or an unindexed BAM (the first N
 * records that reach process_record, util.rs:519-575), --ignore-index.
 * --num-reads N on an indexed BAM follows the reference's sampling schedule (run_extract_reads, src/extract/util.rs:329-470;
 * SamplingSchedule::from_num_reads + get_record_sampler, src/reads_sampler/sampling_schedule.rs:171-273, 417-438): one sampler per interval
 * of the feeder (with --include-bed: of the BED-optimised reference records), then the records without coordinates; rows in interval order.
 * --ignore-implicit drops the inferred calls where the reference does: in its interval path (an index, no --ignore-index; util.rs:413-419) — its
 * serial scan takes the flag and never looks at it.  --exclude-bed drops the rows whose reference position and reference mod strand the BED
 * lists (ReferencePositionFilter::keep, util.rs:44-69).  --motif M off / --cpg (with --ref, --mask): the include filter becomes the motif hits
 * over the whole contigs, intersected with --include-bed (load_regions, util.rs:157-277).  --seed goes to the estimate.  --bgzf writes the table as
 * BGZF blocks (SAM spec 4.1) closed by the EOF block. */
int mkp_extract_calls_main(int argc, const char* const* argv, char* errbuf, size_t errbuf_len);

/* ---- BGZF inflate on the device as a call of its own.  On the pileup path the same kernels run inside the device ingest
 * (`mkp_pileup_main` on an indexed BAM: compressed blocks up, inflate + CRC-32 + record cut + MM/ML tokeniser + packing in HBM, a digest
 * back — DESIGN.md §3.6); this entry point exists so that the decoders are tested and measured alone.  Stands in for what htslib does
 * under rust-htslib's IndexedReader (src/pileup/mod.rs:732-743): BGZF blocks (SAM spec 4.1) of raw DEFLATE (RFC 1951) — neither htslib
 * nor zlib is part of the reference checkout, the decoders follow the RFC.  bgzf = n_bytes of whole BGZF blocks in host memory (a file
 * image).  One wave (small launches) or one thread (large ones) decodes a block; every block's CRC32 and ISIZE are checked before the
 * call returns.  *out = the inflated bytes, owned by the ctx until its next inflate call; *kernel_ms (may be NULL) = device time of the
 * decode kernel. */
int mkp_bgzf_inflate(mkp_ctx* ctx, const uint8_t* bgzf, uint64_t n_bytes, const uint8_t** out, uint64_t* out_len, double* kernel_ms);

/* ---- host-side pieces exposed for tests (no device needed):
 * mkp_host_mm_ranks: the packer's MM tokeniser (MmTagInfo::parse, src/mod_bam.rs:909-1000) on one MM string: for every tag its
 *   header (fundamental base A,C,G,T,N = 0..4; strand; mode 0 '?', 1 '.', 2 none; codes as code_repr) and its delta list turned into
 *   the cumulative occurrence ranks sum(d+1)-1 the device consumes (DeltaListConverter::to_positions_specific, 697-733, is rank ->
 *   position).  Returns the number of tags, or a negative status (MKP_E_INVALID = the reference would reject the tag).
 * mkp_host_map_order: iteration order of a small FxHashMap<ModCodeRepr, _> filled in the given insertion order (rustc-hash 1.1 +
 *   hashbrown), which decides probability ties in MultipleThresholdModCaller::call (src/threshold_mod_caller.rs:28-63). */
typedef struct { uint8_t base, negative_strand, mode, n_codes; uint32_t codes[4]; uint32_t rank_off, n_ranks; } mkp_host_tag;
int mkp_host_mm_ranks(const char* mm, uint32_t l_seq, uint32_t n_ml, mkp_host_tag* tags, uint32_t tags_cap, uint32_t* ranks, uint32_t ranks_cap);
int mkp_host_map_order(const uint32_t* code_reprs, uint32_t n, uint32_t* order_out);

#ifdef __cplusplus
}
#endif
#endif /* MKPILEUP_H */

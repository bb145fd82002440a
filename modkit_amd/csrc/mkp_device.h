// Structures shared by the host packer and the gfx950 kernels.
// Data layout in HBM for one shard (a reference window of one contig):
//   MkpReadHdr  hdr[n_reads]        64 B each, coordinate order
//   uint32      cigar[]             BAM cigar words (len<<4|op)
//   uint2       chunk_pfx[]         per read, per 64 CIGAR ops: query / reference offsets at the chunk start (the host walks the
//                                   CIGAR anyway to get the read's reference span), so a tile starts its walk at the right chunk
//   uint8       seq[]               BAM 4-bit packed bases, each read 4-byte aligned
//   MkpTagRef   tagref[]            one per MM tag of each read
//   uint32      ranks[]             per call: cumulative occurrence index of the tag's fundamental
//                                   base in the as-sequenced read (Σ(delta+1)-1, mod_bam.rs:697-733),
//                                   or the absolute forward position for `N` tags (735-767)
//   uint8       ml[]                ML qualities (bytes of the B:C array)
//   MkpLayout   layout[]            interned MM header structure + caller tables (host-built)
//   uint32      read_ids[]          reads by decode kernel class (FAST 1 tag | FAST 2 tags | general), longest first in a class
//   MkpEvent    events[]            written by the decode kernels, read by mkp_pileup_tiles
//   MkpReadOut  readout[n_reads]    per-read decode summary
//   MkpTile     tiles[n_tiles]      reference range + read range of every tile, genome order (host tile planner)
//   uint32      slotbm[]            focus runs only: one bit per reference position of the window (+64 positions of margin each
//                                   side), set where the position is in focus — the positions that get a tally column ("slot")
//   MkpRowsDev  rows                SoA row buffers (per tile, then gathered into genome order)
#pragma once
#include <stdint.h>

#define MKP_MAX_TAGS 8        // MM tags per read
#define MKP_KMAX 4            // distinct mod codes per (mod strand, base) group
#define MKP_MAX_MEMBERS 4     // tags per (mod strand, base) group
#define MKP_MAX_SLOTS 12      // distinct (primary base, code) pairs per run
#define MKP_MAX_COUNTERS (6 + 4 + MKP_MAX_SLOTS)
#define MKP_PAT_INFERRED 16   // pattern index of the all-inferred ('.' mode) case
#define MKP_HALO 16           // tile halo for strand combining (motif length <= 16)

// counter ids inside one strand tally (Tally, pileup/mod.rs:167-174)
#define MKP_C_NC 0            // NoCall(base) 0..3
#define MKP_C_DEL 4
#define MKP_C_FAIL 5
#define MKP_C_CAN 6           // + index of primary base among the run's primary bases; MOD ids follow

struct MkpReadHdr {
  int32_t ref_start, ref_end;
  uint32_t l_seq, n_cigar;
  uint32_t cigar_off;   // index into cigar[]
  uint32_t seq_off;     // byte offset into seq[]
  uint32_t tag_off;     // index into tagref[]
  uint16_t n_tags;
  uint16_t layout;
  uint32_t flags;       // bit0 reverse; bit1 host-detected tag error (coverage-only read); bit2 MKP_RF_SUMERR; bits 8.. partition key id (0 = ungrouped)
  uint32_t event_off;   // index into events[]
  uint32_t event_cap;
  uint32_t chunk_off;   // index into chunk_pfx[]: one {query offset, reference offset} per 64 CIGAR ops of this read
  // focus runs (slot pipeline, mkp_slots.hip): the read's focus positions ("slots") are the global slots [gs0, gs0 + n_sl) of
  // slot_pos[]; its features go to cov[cov_off .. cov_off + n_sl) (4-byte aligned).  Filled by the host planner (make_resident).
  uint32_t gs0, n_sl, cov_off;
  uint32_t pad;
};
#define MKP_RF_REVERSE 1u
#define MKP_RF_BAD 2u
#define MKP_RF_SUMERR 4u   // host planner: some call's probabilities over the read's two tags add up to more than 1.01 (combine_checked, mod_bam.rs:629-656)
#define MKP_RF_KEY_SHIFT 8
#define MKP_NO_KEY_FILTER 0xffffffffu

struct MkpTagRef { uint32_t rank_off, n, ml_off, pad; };

struct MkpEvent {  // the packed (ref_pos, strand, class, base) event of BASELINE.json's north_star
  uint32_t pos;
  uint32_t info;   // [0:7] counter id of the call class, [8] tally strand of the call,
                   // [9:10] read base, [11] alignment strand, [12] decrement NoCall(read base)
};

struct MkpReadOut { uint32_t n_events; uint32_t ok; uint32_t obs[2]; };  // obs: slot bitmask per tally strand

struct MkpTagDesc { uint8_t fb /*0..3, 4 = N*/, neg, mode /*0 ? 1 . 2 default*/, n_codes; };

struct MkpGroupDesc {          // one (mod strand σ, read base b) group; index σ*4+b.  128 bytes, dword-packed so a
                               // lane fetches everything it needs with a handful of independent LDS reads.
  uint32_t misc;               // [0:2] n_members, [3:6] implicit members (modes '.'/none, fb != N), [7:9] collapse local+1 (0 = none),
                               // [10:14] counter id of Canonical(threshold_base), [16:17] threshold_base, [20:22] n_codes
  uint32_t slots;              // 4 x 8 bit: local code -> global slot
  uint32_t cids;               // 4 x 8 bit: local code -> counter id of Modified(code)
  uint32_t member_tags;        // 4 x 4 bit: member -> tag index
  float thr_mod[MKP_KMAX];     // pass threshold per local code (threshold_mod_caller.rs:36-43)
  float thr_can;
  uint32_t pad[3];
  uint32_t pat[20];            // per hit pattern (1..15, 16 = all-inferred): [0:2] n_pre, [3:5] n_post,
                               // [8:15] FxHashMap iteration order before collapse (4 x 2 bit local idx), [16:23] after collapse
};
#define MKP_G_NMEM(m) ((m) & 7u)
#define MKP_G_IMPL(m) (((m) >> 3) & 15u)
#define MKP_G_COLL(m) ((int)(((m) >> 7) & 7u) - 1)
#define MKP_G_CIDCAN(m) (((m) >> 10) & 31u)
#define MKP_G_TB(m) (((m) >> 16) & 3u)

struct MkpLayout {
  uint8_t n_tags;
  uint8_t default_mask;   // tags whose mode is DefaultImplicitUnmodified
  uint8_t fast;           // 1: all tags share one specific base and mod strand and no code is listed twice (single group, distinct codes)
                          // 2: duplex — two such groups on different bases (`C+h?;C+m?;G-h?;G-m?`): the first `pad` tags form one, the rest the other
  uint8_t pad;            // fast == 2: tags in the first group
  MkpTagDesc tags[MKP_MAX_TAGS];
  uint32_t tagmap[MKP_MAX_TAGS][4];  // per (tag, read base): [0:3] member index in its group, [4+4i..] local code idx of the tag's i-th code
  uint32_t pad2[7];                  // groups start 16-byte aligned (offset 192)
  MkpGroupDesc groups[8];
};

#define MKP_LAYOUT_DWORDS 304
#define MKP_LAYOUT_GROUP_DW 48
#ifdef __cplusplus
static_assert(sizeof(MkpReadHdr) == 64, "read header is 16 dwords");
static_assert(sizeof(MkpGroupDesc) == 128, "group desc is 32 dwords");
static_assert(sizeof(MkpLayout) == 4 * MKP_LAYOUT_DWORDS, "layout is 304 dwords");
#endif

struct MkpSlot { uint32_t code_repr; uint8_t pb; uint8_t cid; uint8_t can_cid; uint8_t pad; };

struct MkpCombo {  // == mkp_motif_combo
  uint8_t n_pos, n_neg;
  uint8_t pos_ids[4], neg_ids[4];
  int8_t pos_delta[4];
  uint8_t pad[2];
};

// (Every kernel that takes this block — by value or through LDS — is compiled against its size: in round 5 one more dword here re-rolled
//  the register allocation of mkp_pileup_tiles into a build 8 % slower on the C2 workload.  After changing it, A/B the accumulate kernels
//  against the previous build on one box: tools/dbg/build_variant.sh + tools/dbg/r5_ab_lib.sh.)
struct MkpRunParams {
  // reference window
  int32_t win_start, win_end;
  uint32_t slot_cap, focus_words;   // S: tally columns per tile; W: bitmap words per tile (0 = no focus: slot = position)
  // counters
  uint32_t n_counters;      // per strand tally
  uint32_t n_slots;
  uint32_t n_pb;            // distinct primary bases with CAN counters
  uint32_t numeric_mode;    // 0 passthrough / 2 collapse (same row logic), 1 combine
  uint32_t combine_strands;
  uint32_t edge_filter, edge_start, edge_end, edge_inverted;
  uint32_t force_allow;
  uint32_t max_depth;
  uint32_t has_focus;
  uint32_t n_combos;
  uint32_t row_capacity;
  uint32_t sample_mode;     // 1: threshold sampling pass — emit argmax probabilities instead of call events
  uint32_t only_mapped;     // sampling: keep only calls with an aligned reference position
  uint32_t debug_skip;      // ablation only (-DMKP_DEBUG builds, env MKP_DEBUG_SKIP): 1 depth walk, 2 events, 4 rows
  uint8_t pb_of_can[4];     // CAN counter k -> primary base
  uint8_t can_of_pb[4];     // primary base -> CAN counter k or 0xff
  uint8_t slot_order[MKP_MAX_SLOTS];   // slots sorted by (code_repr, pb)
  MkpSlot slots[MKP_MAX_SLOTS];
  // pileup-hemi (duplex.rs): per '+' motif position and primary base, one counter per (positive-strand call, negative-strand call)
  // pattern.  A pattern element is 0 for a canonical call, 1.. for the base's mod codes in DuplexModCodeRepr order.
  uint32_t hemi;                          // 1: mkp_pileup_tiles_hemi tallies
  int32_t hemi_off;                       // MotifInfo::negative_strand_position: partner position = position + hemi_off
  uint32_t hemi_counters;                 // counters per tally column (MKP_H_* layout below)
  uint8_t hemi_pat_base[4];               // primary base -> its first pattern counter (0xff: the base has no calls in this run)
  uint8_t hemi_nel[4];                    // primary base -> pattern elements (1 + mod codes; 2 with --combine-mods)
  uint8_t hemi_el[MKP_MAX_COUNTERS + 2];  // call-event counter id -> pattern element; 0xff = Filtered
  uint32_t readout_b_off;                 // duplex reads decoded one group per wave: the second group's summary sits at readout[readout_b_off + read]
  uint32_t slot_stream;                   // 1: focus run on the slot pipeline (feature stream + mkp_pileup_stream)
};
// pileup-hemi counters of one tally column: NoCall(base) 0..3, deletions, Filtered(base) 5..8, then the pattern blocks
#define MKP_H_NC 0
#define MKP_H_DEL 4
#define MKP_H_FAIL 5
#define MKP_H_PAT 9
#define MKP_H_MAX_COUNTERS 64

// One tile of mkp_pileup_tiles: rows are emitted for reference positions [r0, r1) (inside the shard window); tallies are kept
// for the slots of [r0 - MKP_HALO, r1 + MKP_HALO) (strand combining reads a partner up to MKP_HALO away); reads [first, last)
// are the candidates overlapping that range.  A "slot" is a position that owns a tally column in LDS: every position of
// the range when the run has no focus, the focus positions only (--cpg / --motif / --include-bed) when it has.
struct MkpTile { int32_t r0, r1; uint32_t first, last; };
#define MKP_SLOTBM_MARGIN 64   // slot bitmap origin = win_start - MKP_SLOTBM_MARGIN

// mkp_pileup_tiles geometry: 16 waves per tile.  Dynamic LDS, in dwords:
//   tallies  [words_per_slot][S]            S = slot capacity of a tile (multiple of 64), 16-bit-packed strand tallies
//   focus runs: bm[W] + pfx[W] (slot bitmap of the tile's reference range and its running popcount), fpos[S] (slot -> position)
//   per wave: an op-start bitmap over the tile's slots (even number of dwords, 2 spare for the 64-bit window read) and a
//   64 x 8-byte compaction buffer
#ifndef MKP_PILEUP_THREADS
#define MKP_PILEUP_THREADS 1024
#endif
#define MKP_PILEUP_WAVE_SCRATCH 128
#define MKP_PILEUP_BM_WORDS(S) (((((S) + 31u) >> 5) + 3u) & ~1u)
// per-wave scratch: dense kernel = op-start bitmap + compaction buffer; focus kernel = one word per slot of a read's visit
// (the packed query index / kind the CIGAR phase leaves for the SEQ phase)
#define MKP_PILEUP_WAVE_WORDS(S, focus_words) ((focus_words) && (S) > MKP_PILEUP_BM_WORDS(S) + MKP_PILEUP_WAVE_SCRATCH ? (S) : MKP_PILEUP_BM_WORDS(S) + MKP_PILEUP_WAVE_SCRATCH)
#define MKP_PILEUP_LDS_WORDS(words_per_slot, S, focus_words) ((words_per_slot) * (S) + ((focus_words) ? 2u * (focus_words) + (S) : 0u) + (MKP_PILEUP_THREADS / 64) * MKP_PILEUP_WAVE_WORDS(S, focus_words))

// ---- slot pipeline (focus runs: --cpg / --motif / --include-bed; mkp_slots.hip) -------------------------------------------
// The decode side leaves one FEATURE BYTE per (read, focus position in the read's reference span) — position-implicit: byte k
// of a read belongs to global slot gs0 + k — and one MkpVisit per read; mkp_pileup_stream is then a pure histogram of that
// stream into LDS tallies.  Feature byte = what FeatureVector::add_feature receives for this alignment at this column
// (pileup/mod.rs:783-939): [0:4] counter id (MKP_C_*), [5] tally strand, [6:7] the read base as tallied (so that a call of a
// record that later fails can be counted as NoCall(base)).
#ifndef MKP_SLOT_WB
// mkp_decode_slots*: stored bases per base window; longer reads take the *_long instances.  13 312 bases = 5 052 bytes of LDS per wave: eight
// 4-wave workgroups per CU (16 384: six), which — together with the 80-SGPR cap of the short-read kernel, see mkp_slots.hip — is what lets the
// hardware keep eight waves per SIMD resident (round 6: 0.75 -> 0.70 ms on C3)
#define MKP_SLOT_WB 13312u
#endif
#define MKP_STREAM_ROWMAP_WORDS 2048u   // mkp_pileup_stream: rows of a tile placed per emission round (a dword of LDS each)
#define MKP_FB_NONE 0xffu    // the read is not in this column (ref-skip)
#define MKP_FB_BLANK 0xfeu   // in the column, no feature (non-ACGT base: pileup/mod.rs:864-874)
struct MkpVisit {            // 32 B, written by the decode / cover kernels, read by mkp_pileup_stream
  uint32_t gs0, n_sl, cov_off;
  uint32_t flags;            // bit0 record yielded calls (observed-code masks valid), bit1 alignment strand, bit2 stream holds NONE bytes,
                             // bits 8.. partition key id
  uint32_t obs0, obs1;       // observed-code slot masks per tally strand (read_cache.rs:171-194)
  uint32_t over_off, n_over; // second features on one column (pos_call and neg_call at one base): {global slot, feature byte} pairs in events[]
};
// what mkp_decode_slots* needs of a layout whose tags form one explicit-mode group (decode class SPARSE), resolved by the host from
// the MkpLayout so that a wave gets it with one scalar load: the caller's walk over a call's map in iteration order
struct MkpFusedDesc {        // 64 B
  uint32_t misc;             // [0:1] fundamental base, [2] mod strand, [3:5] codes the caller sees (n_post), [6] / [7] the integer form of the caller is exact
                             // without / with the collapse, [8:15] counter of Canonical, [16:31] observed-code slot mask
  uint32_t it_cid;           // 4 x 8 bit: counter of Modified(i-th code)
  uint32_t it_src;           // 4 x 4 bit: where the i-th code's ML byte sits: [0] tag, [1:3] index among the tag's codes
  uint32_t nc;               // codes per call of tag 0 | tag 1 << 8 (the tags' ML strides)
  float it_thr[4];           // pass threshold of the i-th code
  float thr_can;
  // --ignore / --preset traditional: ReDistribute(x) (BaseModProbs::into_collapsed, mod_bam.rs:558-600) — when the map holds x, every
  // other code gets p_x / n_other added (n_other = number of codes before the collapse; x is no longer among the it_* entries)
  uint32_t col;              // [0] the map holds x, [1:4] where x's ML byte sits (tag | index << 1), [5:6] log2(n_other) when it is 1, 2 or 4
  float n_other;
  // The caller in integers (round 6).  (q + 0.5) / 256 is a multiple of 2^-9 and, when n_other is a power of two, so is every value the
  // f32 walk forms — shares, sums, 1 - sum — a multiple of 2^-11 that f32 holds exactly: `p >= threshold` is `p * 2048 >= i_thr` with
  // i_thr = the least integer T such that T / 2048 >= threshold (INT32_MAX for a NaN threshold: never passed).
  int32_t i_can;
  int32_t i_thr[4];          // (16-byte aligned: one LDS vector read per slot batch)
};
static_assert(sizeof(MkpFusedDesc) == 64, "MkpFusedDesc is one 64-byte scalar load");
// one read as mkp_decode_slots* takes it: the header and tag fields the kernel needs, in launch order (longest reads first), so that a
// wave starts from ONE 64-byte scalar load instead of the chain read id -> header -> tag table
struct MkpWork {             // 64 B
  int32_t ref_start; uint32_t l_seq, n_cigar, cigar_off, seq_off, flags, gs0, n_sl, cov_off;
  uint16_t n_tags, layout;
  uint32_t rank_off, n_calls, ml_off0, ml_off1;   // the shared rank list, the tags' ML bytes
  uint32_t rid, pad;
};
// Records that share a read NAME inside one interval of the reference's grid (unmarked duplicates, mates, split reads that kept the primary flag).
// The reference keeps ONE cache entry per name and interval (ReadCache, read_cache.rs:24-43): the record asked about first — the first focus
// column of the interval that holds a record of the name, file order within a column — is parsed; every later record of the name is answered
// from THAT record's call map, looked up by reference position and the asking record's own read base (get_mod_call, 232-297).  The host planner
// works out, per interval, which record owns the name (make_resident); a record that is answered from another one in some interval is a
// CONSUMER: mkp_dup_events rebuilds its event list — its own events where it owns the name, the owner's events elsewhere, kept where its own
// alignment shows the call's base — and the accumulate kernels then take that list as the record's own.
// positions [p_lo, p_hi) of one consumer, ascending
struct MkpDupSeg { int32_t p_lo, p_hi; uint32_t src_off /* the owner's OWN event slice */, owner; };
struct MkpDupCons {
  uint32_t rid, seg_off, n_seg;
  uint32_t own_off;    // the record's own event slice (what its decode kernel writes)
  uint32_t eff_off, eff_cap;   // the rebuilt list
  uint32_t out_n, out_ok, out_obs0, out_obs1;   // written by mkp_dup_events, moved into the record's header / summary by mkp_dup_apply
  uint32_t pad[6];
};
#define ERR_DUP_MIXED 8u   // device error bit: the records a consumer is answered from disagree in status or observed codes (refused, not approximated)
#define MKP_VF_OK 1u
#define MKP_VF_REV 2u
#define MKP_VF_GAPS 4u
// one tile of mkp_pileup_stream: rows for the global slots [g0, g1) (positions [r0, r1)), tally columns for [gh0, gh1) (+- MKP_HALO
// positions for strand combining), candidate reads [first, last)
struct MkpSTile { uint32_t g0, g1, gh0, gh1; int32_t r0, r1; uint32_t first, last; };

struct MkpRowsDev {  // SoA row buffers (44 B / row)
  uint32_t* pos; uint32_t* info; uint32_t* code;
  uint32_t* n_valid; uint32_t* n_mod; uint32_t* n_can; uint32_t* n_other;
  uint32_t* n_del; uint32_t* n_fail; uint32_t* n_diff; uint32_t* n_nocall;
};
#ifdef __cplusplus
static_assert(sizeof(MkpWork) == 64, "work record is 16 dwords");
static_assert(sizeof(MkpFusedDesc) == 64 && sizeof(MkpVisit) == 32, "slot pipeline records");
#endif

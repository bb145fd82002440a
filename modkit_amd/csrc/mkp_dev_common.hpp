// Device-side helpers shared by the gfx950 kernels of libmkpileup (wave64): lane / DPP primitives, packed-base helpers and the
// f32 threshold caller.  Included by mkp_kernels.hip and mkp_slots.hip only.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <type_traits>

#include "mkp_device.h"

#define ERR_EVENT_CAP 1u
#define ERR_ROW_CAP 2u
#define ERR_DEPTH 4u

__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63u); }
__device__ __forceinline__ unsigned long long lanemask_le() { int l = lane_id(); return l == 63 ? ~0ull : ((1ull << (l + 1)) - 1ull); }
__device__ __forceinline__ unsigned long long lanemask_lt() { return (1ull << lane_id()) - 1ull; }

// Inclusive prefix sum over the wave: four row_shr steps inside each row of 16 lanes, then row_bcast:15 / row_bcast:31
// carry the row totals across (lanes without a source add 0).  Six v_add_u32_dpp, no LDS traffic.
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v) {
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111 /*row_shr:1*/, 0xf, 0xf, false);
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112 /*row_shr:2*/, 0xf, 0xf, false);
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114 /*row_shr:4*/, 0xf, 0xf, false);
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118 /*row_shr:8*/, 0xf, 0xf, false);
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142 /*row_bcast:15*/, 0xa, 0xf, false);
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143 /*row_bcast:31*/, 0xc, 0xf, false);
  return v;
}
__device__ __forceinline__ uint32_t wave_or(uint32_t v) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v |= __shfl_xor(v, d, 64);
  return v;
}

// BAM 4-bit code -> A,C,G,T = 0..3, anything else -1 (DnaBase::parse, mod_base_code.rs:188-196)
// one-hot nibble -> bit index; 0 -> -1
__device__ __forceinline__ int nib2base(uint32_t n) { const int x = (int)__ffs((int)n) - 1; return (n & (n - 1u)) ? -1 : x; }
__device__ __forceinline__ uint32_t seq_nibble(const uint8_t* __restrict__ s, uint32_t q) {
  uint32_t b = s[q >> 1];
  return (q & 1u) ? (b & 15u) : (b >> 4);
}
// CIGAR op classes as bit tables over the op code (MIDNSHP=X = 0..8): no branches
__device__ __forceinline__ bool op_consumes_query(uint32_t op) { return (0x193u >> op) & 1u; }  // M I S = X
__device__ __forceinline__ bool op_consumes_ref(uint32_t op) { return (0x18du >> op) & 1u; }    // M D N = X
__device__ __forceinline__ bool op_is_match(uint32_t op) { return (0x181u >> op) & 1u; }        // M = X

// number of lanes whose (non-decreasing) inclusive prefix `incl` is <= j  == index of the op holding element j
__device__ __forceinline__ int find_op(uint32_t incl, uint32_t j) {
  int idx = 0;
#pragma unroll
  for (int step = 32; step >= 1; step >>= 1) {
    uint32_t v = __shfl(incl, idx + step - 1, 64);
    if (v <= j) idx += step;
  }
  return idx;
}

__device__ __forceinline__ uint32_t sel4(const uint32_t* a, int x) { return x == 0 ? a[0] : x == 1 ? a[1] : x == 2 ? a[2] : a[3]; }
__device__ __forceinline__ unsigned long long sel4b(const unsigned long long* a, int x) { return x == 0 ? a[0] : x == 1 ? a[1] : x == 2 ? a[2] : a[3];
  }

__device__ __forceinline__ int find_rank(const uint32_t* __restrict__ a, uint32_t n, uint32_t key) {
  uint32_t lo = 0, hi = n;
  while (lo < hi) {
    uint32_t mid = (lo + hi) >> 1;
    if (a[mid] < key) lo = mid + 1; else hi = mid;
  }
  return (lo < n && a[lo] == key) ? (int)lo : -1;
}

// Four f32 values addressed by a small index, kept as named scalars: a `float[4]` indexed by a lane-varying value is
// demoted to scratch memory by the compiler (a select chain over array elements becomes an indexed load), which put
// scratch loads into the per-call path.  With scalars the selects stay v_cndmask.
struct F4 { float v0, v1, v2, v3; };
// k is a compile-time constant at every call site (unrolled loops)
__device__ __forceinline__ float& at(F4& f, int k) { return k == 0 ? f.v0 : k == 1 ? f.v1 : k == 2 ? f.v2 : f.v3; }
__device__ __forceinline__ float at(const F4& f, int k) { return k == 0 ? f.v0 : k == 1 ? f.v1 : k == 2 ? f.v2 : f.v3; }
__device__ __forceinline__ float getk(const F4& p, int k) {   // OR of masked bit patterns: cannot be folded into an indexed (scratch) load
  const uint32_t r = (k == 0 ? __float_as_uint(p.v0) : 0u) | (k == 1 ? __float_as_uint(p.v1) : 0u) | (k == 2 ? __float_as_uint(p.v2) : 0u) | (k == 3
      ? __float_as_uint(p.v3) : 0u);
  return __uint_as_float(r);
}
// p[kk] = cond ? v : p[kk], as bit-field inserts under an all-ones / all-zeros mask (again: nothing the compiler can turn into an indexed store)
__device__ __forceinline__ float bsel(bool c, float a, float b) { const uint32_t m = c ? 0xffffffffu : 0u;
  return __uint_as_float((__float_as_uint(a) & m) | (__float_as_uint(b) & ~m)); }
__device__ __forceinline__ void setk(F4& p, uint32_t kk, bool cond, float v) {
  p.v0 = bsel(cond && kk == 0u, v, p.v0); p.v1 = bsel(cond && kk == 1u, v, p.v1); p.v2 = bsel(cond && kk == 2u, v, p.v2);
    p.v3 = bsel(cond && kk == 3u, v, p.v3);
}
__device__ __forceinline__ void addk(F4& p, int k, float v) { p.v0 = k == 0 ? p.v0 + v : p.v0; p.v1 = k == 1 ? p.v1 + v : p.v1;
  p.v2 = k == 2 ? p.v2 + v : p.v2; p.v3 = k == 3 ? p.v3 + v : p.v3; }

// ----------------------------------------------------------------------------------------------
// One (mod strand, base) group descriptor as the lane sees it (fetched from the LDS copy of the layout).
struct GroupRegs { uint32_t misc, slots, cids, member_tags; F4 thr; float thr_can; };
__device__ __forceinline__ GroupRegs load_group(const uint32_t* g) {
  GroupRegs r;
  const uint4 a = *reinterpret_cast<const uint4*>(g);
  const float4 t = *reinterpret_cast<const float4*>(g + 4);
  r.misc = a.x; r.slots = a.y; r.cids = a.z; r.member_tags = a.w;
  r.thr.v0 = t.x; r.thr.v1 = t.y; r.thr.v2 = t.z; r.thr.v3 = t.w;
  r.thr_can = __uint_as_float(g[8]);
  return r;
}

// ReDistribute collapse (BaseModProbs::into_collapsed, mod_bam.rs:558-600) in the map's iteration order.
__device__ __forceinline__ void collapse_redistribute(const GroupRegs& g, uint32_t pv, F4& pk, int kmax) {
  const int n_pre = (int)(pv & 7u);
  const int x = MKP_G_COLL(g.misc);
  bool present = false;
#pragma unroll
  for (int i = 0; i < MKP_KMAX; i++) { if (i >= kmax) break; present |= (i < n_pre) && ((int)((pv >> (8 + 2 * i)) & 3u) == x); }
  const float marginal = present ? getk(pk, x) : 0.0f;
  const float n_other = (float)(present ? n_pre : n_pre + 1);  // other_mods.len() + 1
  const float redistribute = marginal / n_other;
#pragma unroll
  for (int i = 0; i < MKP_KMAX; i++) { if (i >= kmax) break; const int kq = (int)((pv >> (8 + 2 * i)) & 3u);
    addk(pk, (i < n_pre && kq != x) ? kq : -1, redistribute); }
}

// BaseModProbs -> BaseModCall: MultipleThresholdModCaller::call (threshold_mod_caller.rs:28-63).
// Returns 0 Filtered, 1 Canonical, 2+k Modified(local code k); *obs gets the slots of the codes in the map the
// caller sees (read_cache.rs:171-179).  pv = the group's entry for this hit pattern.
// kmax: wave-uniform bound on the number of codes in the map (MKP_KMAX when the group differs per lane)
__device__ __forceinline__ int call_group(const GroupRegs& g, uint32_t pv, F4& pk, bool collapse, uint32_t* obs, int kmax = MKP_KMAX) {
  if (collapse) collapse_redistribute(g, pv, pk, kmax);
  const int n_post = (int)((pv >> 3) & 7u);
  int best = 0;
  float best_p = 0.0f, s = 0.0f;
  uint32_t ob = 0;
#pragma unroll
  for (int i = 0; i < MKP_KMAX; i++) {   // predicated, uniform trip count
    if (i >= kmax) break;
    const bool valid = i < n_post;
    const int kq = (int)((pv >> (16 + 2 * i)) & 3u);
    const float p = getk(pk, kq);
    ob |= valid ? (1u << ((g.slots >> (8 * kq)) & 0xffu)) : 0u;
    s = valid ? s + p : s;  // probs.values().sum() in map order
    const bool take = valid && p >= getk(g.thr, kq) && (best == 0 || !(p < best_p));  // Iterator::max keeps the last maximum
    best = take ? 2 + kq : best; best_p = take ? p : best_p;
  }
  const float pc = 1.0f - s;  // canonical_prob, pushed last
  const bool takec = pc >= g.thr_can && (best == 0 || !(pc < best_p));
  best = takec ? 1 : best;
  *obs |= ob;
  return best;
}

// Threshold sampling: value of BaseModProbs::argmax_base_mod_call after the optional collapse
// (mod_bam.rs:489-505; read_ids_to_base_mod_probs.rs:67-101, 324-328).
__device__ __forceinline__ float argmax_group(const GroupRegs& g, uint32_t pv, F4& pk, bool collapse, int kmax = MKP_KMAX) {
  if (collapse) collapse_redistribute(g, pv, pk, kmax);
  const int n_post = (int)((pv >> 3) & 7u);
  float s = 0.0f, best = 0.0f; bool have = false;
#pragma unroll
  for (int i = 0; i < MKP_KMAX; i++) {
    if (i >= kmax) break;
    const bool valid = i < n_post;
    const float p = getk(pk, (int)((pv >> (16 + 2 * i)) & 3u));
    s = valid ? s + p : s;
    const bool take = valid && (!have || !(p < best));
    best = take ? p : best; have = have || valid;
  }
  const float can = 1.0f - s;
  return (have && best > can) ? best : can;
}

// `modkit summary` (sampled_reads_to_summary, src/summarize.rs:117-262): per sampled call the thresholded call and the argmax call.
// Returns the sample "event" info: [0:1] canonical base, [4:7] thresholded class, [8:11] argmax class; class 0 = Filtered,
// 1 = Canonical, 2 + s = Modified(code of global slot s).  *obs gets the slots of the codes in the map (observed_mods).
// *amax (optional): the probability of the argmax call (`extract calls`: call_prob)
__device__ __forceinline__ uint32_t summary_info(const GroupRegs& g, uint32_t pv, F4& pk, bool collapse, uint32_t* obs, int kmax = MKP_KMAX,
    float* amax = nullptr) {
  if (collapse) collapse_redistribute(g, pv, pk, kmax);
  const int n_post = (int)((pv >> 3) & 7u);
  float s = 0.0f, best = 0.0f; bool have = false; int bk = 0;
#pragma unroll
  for (int i = 0; i < MKP_KMAX; i++) {   // argmax_base_mod_call (mod_bam.rs:489-505): max_by keeps the last maximum
    if (i >= kmax) break;
    const bool valid = i < n_post;
    const int kq = (int)((pv >> (16 + 2 * i)) & 3u);
    const float p = getk(pk, kq);
    s = valid ? s + p : s;
    const bool take = valid && (!have || !(p < best));
    best = take ? p : best; bk = take ? kq : bk; have = have || valid;
  }
  const float can = 1.0f - s;
  const uint32_t arg_cls = (have && best > can) ? 2u + ((g.slots >> (8 * bk)) & 0xffu) : 1u;
  if (amax) *amax = (have && best > can) ? best : can;
  const int cls = call_group(g, pv, pk, false, obs, kmax);   // on the collapsed map
  const uint32_t thr_cls = cls < 2 ? (uint32_t)cls : 2u + ((g.slots >> (8 * (cls - 2))) & 0xffu);
  return MKP_G_TB(g.misc) | (thr_cls << 4) | (arg_cls << 8);
}

// Per (mod strand) BaseModProbs under construction at one read position.
struct GState { F4 pk; uint32_t H, setmask; };

// combine_positions_to_probs for one more tag at this position (mod_bam.rs:1037-1054, 629-656)
__device__ __forceinline__ bool merge_tag(GState& S, const F4& ts, uint32_t seen, uint32_t mi) {
  bool bad = false;
#pragma unroll
  for (int k = 0; k < MKP_KMAX; k++) if (seen & (1u << k)) { if (S.setmask & (1u << k)) at(S.pk, k) = at(S.pk, k) + at(ts, k);
    else at(S.pk, k) = at(ts, k);
    }
  S.setmask |= seen;
  if (S.H) {
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < MKP_KMAX; k++) if (S.setmask & (1u << k)) s = s + at(S.pk, k);
    if (s > 1.01f) bad = true;
  }
  S.H |= 1u << mi;
  return bad;
}

// index of the lane holding `key` among the wave's sorted entries `e` (or 64): 6 bpermutes, no memory traffic
__device__ __forceinline__ int find_sorted(uint32_t e, uint32_t key) {
  int idx = 0;
#pragma unroll
  for (int step = 32; step >= 1; step >>= 1) {
    uint32_t v = __shfl(e, idx + step - 1, 64);
    if (v < key) idx += step;
  }
  return idx;
}

// ---- packed-base helpers: a dword of BAM SEQ holds 8 bases, high nibble of each byte first.
// linearize() swaps the nibbles of every byte so that base i of the dword sits at bits [4i, 4i+4).
__device__ __forceinline__ uint32_t linearize(uint32_t x) { return ((x & 0x0f0f0f0fu) << 4) | ((x >> 4) & 0x0f0f0f0fu); }
// 8-bit mask (bit i = base i of the dword) of the nibbles equal to the BAM code of base k (A,C,G,T = 1,2,4,8)
__device__ __forceinline__ uint32_t match8(uint32_t xl, int k) {
  uint32_t t = xl ^ (0x11111111u << k);
  t |= t >> 1; t |= t >> 2;
  const uint32_t m = (t & 0x11111111u) ^ 0x11111111u;          // bit 4i set where nibble i matches
  return ((((m | (m >> 3)) & 0x03030303u) * 0x01041040u) >> 24);  // gather bits 4i -> i
}
// position of the r-th (0-based) set bit of an 8-bit mask
__device__ __forceinline__ uint32_t select8(uint32_t m, uint32_t r) {
  for (uint32_t k = 0; k < r; k++) m &= m - 1u;
  return (uint32_t)__ffs((int)m) - 1u;
}

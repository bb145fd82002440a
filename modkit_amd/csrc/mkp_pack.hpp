// Host packer for libmkpileup: turns alignment records (mkp_record = bam1_t view) into the SoA
// shard layout of mkp_device.h and builds the per-layout caller tables the decode kernel reads.
//
// What stays on the host: locating the MM/ML/MN aux fields (mod_bam.rs:1388-1470), tokenising the
// MM text into integers (MmTagInfo::parse, mod_bam.rs:909-1000) — written out as cumulative
// occurrence ranks Σ(d+1)-1 so the device needs no scan over the delta list — and interning the
// MM header structure ("layout") so thresholds / hash-map iteration orders are resolved once per
// structure instead of once per call.  Everything per base and per call happens on the GPU.
#pragma once
#include <cmath>
#include <sys/mman.h>
#include <memory>
#include <thread>
#include <unordered_map>

#include "mkp_bam.hpp"
#include "mkp_device.h"

namespace mkp {

// ------------------------------------------------------------------------------------------
// Iteration order of a small FxHashMap<ModCodeRepr, f32> (rustc-hash 1.1 + hashbrown):
// decides ties in MultipleThresholdModCaller::call and the f32 summation order.
class FxOrder {
 public:
  explicit FxOrder(size_t reserve = 0) { if (reserve) grow_to(buckets_for(reserve)); }
  void insert(uint32_t code_repr, int tag) {  // entry(code).or_insert(..)
    for (auto& b : b_) if (b.used && b.code == code_repr) return;
    if (n_ == capacity(b_.size())) grow_to(b_.empty() ? 4 : b_.size() * 2);
    place(code_repr, tag); n_++;
  }
  std::vector<int> order() const { std::vector<int> o; for (auto& b : b_) if (b.used) o.push_back(b.tag); return o; }
  std::vector<uint32_t> codes() const { std::vector<uint32_t> o; for (auto& b : b_) if (b.used) o.push_back(b.code); return o; }
 private:
  struct B { bool used = false; uint32_t code = 0; int tag = 0; };
  std::vector<B> b_; size_t n_ = 0;
  static size_t capacity(size_t nb) { return nb == 0 ? 0 : (nb < 8 ? nb - 1 : nb / 8 * 7); }
  static size_t buckets_for(size_t cap) { if (cap < 4) return 4; if (cap < 8) return 8; size_t adj = cap * 8 / 7, p = 1; while (p < adj) p <<= 1;
    return p; }
  static uint64_t hash(uint32_t code_repr) {
    auto add = [](uint64_t h, uint64_t w) { return (((h << 5) | (h >> 59)) ^ w) * 0x517cc1b727220a95ull; };
    bool chebi = (code_repr & 0x80000000u) != 0;
    return add(add(0, chebi ? 1 : 0), chebi ? (code_repr & 0x7fffffffu) : code_repr);
  }
  void place(uint32_t code, int tag) {
    size_t nb = b_.size(), pos = (size_t)(hash(code) & (nb - 1));
    if (nb > 16) throw Error(MKP_E_UNSUPPORTED, "more than 14 mod codes in one map");
    for (size_t s = 0; s < nb; s++) { B& b = b_[(pos + s) & (nb - 1)]; if (!b.used) { b.used = true; b.code = code; b.tag = tag; return; } }
  }
  void grow_to(size_t nb) { std::vector<B> old; old.swap(b_); b_.assign(nb, B()); for (auto& b : old) if (b.used) place(b.code, b.tag); }
};

struct TagHeader { uint8_t fb; bool neg; uint8_t mode; std::vector<uint32_t> codes; };
struct LayoutHost { std::vector<TagHeader> tags; };

struct CallerCfg {  // mkp_caller, owned copy
  float default_threshold = 0.f; float per_base[4] = {0, 0, 0, 0}; bool has_per_base[4] = {false, false, false, false};
  std::map<uint32_t, float> per_mod;
  uint32_t numeric_mode = 0, collapse_code = 0; bool edge = false; uint32_t edge_start = 0, edge_end = 0; bool edge_inverted = false;
  bool force_allow = false, combine_strands = false; uint32_t max_depth = 8000;
  // `extract calls`: PositionModCalls::to_row asks the caller with the READ base of the call (src/extract/writer.rs:61), where the pileup's
  // read cache asks with the complement for negative-strand tags (read_cache.rs:147-150)
  bool read_base_caller = false;
};

struct SlotTable {
  std::vector<MkpSlot> slots; std::vector<int> can_pbs;  // primary bases having a CAN counter
  int find_slot(int pb, uint32_t code) const {
    for (size_t i = 0; i < slots.size(); i++) if (slots[i].pb == pb && slots[i].code_repr == code) return (int)i;
    return -1; }
  int find_can(int pb) const { for (size_t i = 0; i < can_pbs.size(); i++) if (can_pbs[i] == pb) return (int)i; return -1; }
};

// Everything the kernels need to know about MM header structures under the current caller.
struct LayoutTables {
  std::vector<MkpLayout> dev; SlotTable st; uint32_t n_counters = 0;

  static void group_members(const LayoutHost& L, int sg, int b, std::vector<int>* members, std::vector<uint32_t>* universe) {
    for (size_t t = 0; t < L.tags.size(); t++) {
      const TagHeader& h = L.tags[t];
      if ((int)h.neg != sg) continue;
      if (!(h.fb == 4 || h.fb == b)) continue;
      members->push_back((int)t);
      for (uint32_t c : h.codes) if (std::find(universe->begin(), universe->end(), c) == universe->end()) universe->push_back(c);
    }
  }

  // `used` (optional): layout ids the current shard's reads refer to; the counters / slots and the LDS tile size then depend on
  // the shard only, not on what the packer interned earlier (e.g. for the threshold sampler).  Unused ids get an empty entry.
  void build(const std::vector<LayoutHost>& layouts, const CallerCfg& cc, const std::vector<uint8_t>* used = nullptr) {
    dev.clear(); st = SlotTable();
    const bool collapse = cc.numeric_mode == 2;
    auto is_used = [&](size_t li) { return !used || (li < used->size() && (*used)[li]); };
    // pass 1: slots and CAN counters over every group of every layout
    for (size_t li = 0; li < layouts.size(); li++) for (int sg = 0; sg < 2; sg++) for (int b = 0; b < 4; b++) {
      if (!is_used(li)) continue;
      const LayoutHost& L = layouts[li];
      std::vector<int> mem; std::vector<uint32_t> uni; group_members(L, sg, b, &mem, &uni);
      if (mem.empty()) continue;
      int pb = (sg && !cc.read_base_caller) ? 3 - b : b;  // threshold_base (read_cache.rs:147-150)
      if (st.find_can(pb) < 0) st.can_pbs.push_back(pb);
      for (uint32_t c : uni) { if (collapse && c == cc.collapse_code) continue; if (st.find_slot(pb, c) < 0) { MkpSlot s; memset(&s, 0, sizeof(s));
          s.code_repr = c; s.pb = (uint8_t)pb; st.slots.push_back(s); } }
    }
    if (st.slots.size() > MKP_MAX_SLOTS) throw Error(MKP_E_UNSUPPORTED,
        "more than " + std::to_string(MKP_MAX_SLOTS) + " distinct (base, mod code) pairs in one run");
    std::sort(st.can_pbs.begin(), st.can_pbs.end());
    n_counters = 6 + (uint32_t)st.can_pbs.size() + (uint32_t)st.slots.size();
    for (size_t i = 0; i < st.slots.size(); i++) { st.slots[i].cid = (uint8_t)(6 + st.can_pbs.size() + i);
      st.slots[i].can_cid = (uint8_t)(6 + st.find_can(st.slots[i].pb)); }
    // pass 2: per-layout tables
    for (size_t li = 0; li < layouts.size(); li++) {
      const LayoutHost& L = layouts[li];
      MkpLayout D; memset(&D, 0, sizeof(D));
      if (!is_used(li)) { dev.push_back(D); continue; }
      D.n_tags = (uint8_t)L.tags.size();
      { bool fast = !L.tags.empty(); std::vector<uint32_t> seen_codes;
        for (auto& th : L.tags) { if (th.fb == 4 || th.fb != L.tags[0].fb || th.neg != L.tags[0].neg || th.codes.empty()) fast = false;
          for (uint32_t c : th.codes) { if (std::find(seen_codes.begin(), seen_codes.end(), c) != seen_codes.end()) fast = false;
            seen_codes.push_back(c); } }
        D.fast = fast ? 1 : 0;
        // duplex: a leading run of tags on one (strand, base) and the rest on another one, different base, at most two tags each and no
        // code listed twice inside a group (`C+h?;C+m?;G-h?;G-m?`, `C+hm?;G-hm?`): each group decodes like a single-group read
        if (!fast && L.tags.size() >= 2 && L.tags.size() <= 4) {
          size_t nA = 1; while (nA < L.tags.size() && L.tags[nA].fb == L.tags[0].fb && L.tags[nA].neg == L.tags[0].neg) nA++;
          bool ok = nA < L.tags.size() && nA <= 2 && L.tags.size() - nA <= 2 && L.tags[0].fb < 4 && L.tags[nA].fb < 4
              && L.tags[nA].fb != L.tags[0].fb;
          for (size_t t = nA; ok && t < L.tags.size(); t++) ok = L.tags[t].fb == L.tags[nA].fb && L.tags[t].neg == L.tags[nA].neg;
          for (size_t g = 0; ok && g < 2; g++) { std::vector<uint32_t> seen; for (size_t t = g ? nA : 0; t < (g ? L.tags.size() : nA); t++) {
              if (L.tags[t].codes.empty()) ok = false;
              for (uint32_t c : L.tags[t].codes) { if (std::find(seen.begin(), seen.end(), c) != seen.end()) ok = false;
                seen.push_back(c); } } }
          if (ok) { D.fast = 2; D.pad = (uint8_t)nA; }
        } }
      for (size_t t = 0; t < L.tags.size(); t++) {
        D.tags[t].fb = L.tags[t].fb; D.tags[t].neg = L.tags[t].neg; D.tags[t].mode = L.tags[t].mode;
          D.tags[t].n_codes = (uint8_t)L.tags[t].codes.size();
        if (L.tags[t].mode == 2) D.default_mask |= (uint8_t)(1u << t);
      }
      for (int sg = 0; sg < 2; sg++) for (int b = 0; b < 4; b++) {
        MkpGroupDesc& G = D.groups[sg * 4 + b];
        std::vector<int> mem; std::vector<uint32_t> uni; group_members(L, sg, b, &mem, &uni);
        if (mem.empty()) continue;
        if (mem.size() > MKP_MAX_MEMBERS) throw Error(MKP_E_UNSUPPORTED, "more than 4 MM tags on one (strand, base)");
        if (uni.size() > MKP_KMAX) throw Error(MKP_E_UNSUPPORTED, "more than 4 mod codes on one (strand, base)");
        int pb = (sg && !cc.read_base_caller) ? 3 - b : b;
        int collapse_local = -1; uint32_t implicit = 0;
        auto local_of = [&](uint32_t c) { return (int)(std::find(uni.begin(), uni.end(), c) - uni.begin()); };
        for (size_t k = 0; k < uni.size(); k++) {
          bool gone = collapse && uni[k] == cc.collapse_code;
          if (gone) { collapse_local = (int)k; G.cids |= (uint32_t)MKP_C_FAIL << (8 * k); }
          else { int s = st.find_slot(pb, uni[k]); G.slots |= (uint32_t)s << (8 * k); G.cids |= (uint32_t)st.slots[(size_t)s].cid << (8 * k); }
          // threshold resolution (threshold_mod_caller.rs:36-43)
          float thr; auto a = cc.per_mod.find(uni[k]);
          if (a != cc.per_mod.end()) thr = a->second;
          else { auto any = cc.per_mod.find((uint32_t)"ACGT"[pb]); if (any != cc.per_mod.end()) thr = any->second;
            else thr = cc.has_per_base[pb] ? cc.per_base[pb] : cc.default_threshold;
            }
          G.thr_mod[k] = thr;
        }
        G.thr_can = cc.has_per_base[pb] ? cc.per_base[pb] : cc.default_threshold;
        for (size_t mi = 0; mi < mem.size(); mi++) {
          const TagHeader& h = L.tags[(size_t)mem[mi]];
          G.member_tags |= (uint32_t)mem[mi] << (4 * mi);
          uint32_t tm = (uint32_t)mi;
          for (size_t i = 0; i < h.codes.size(); i++) tm |= (uint32_t)local_of(h.codes[i]) << (4 + 4 * i);
          D.tagmap[mem[mi]][b] = tm;
          if (h.mode != 0 && h.fb != 4) implicit |= 1u << mi;
        }
        G.misc = (uint32_t)mem.size() | (implicit << 3) | ((uint32_t)(collapse_local + 1) << 7) | ((uint32_t)(6 + st.find_can(pb)) << 10) | ((uint32_t)pb << 16) | ((uint32_t)uni.size() << 20);
        auto fill = [&](int pat, const std::vector<int>& hit_members, bool inferred) {
          // maps as the reference builds them: each tag's own map first (new_init / new_inferred_canonical),
          // then merged into the aggregate in MM order (combine_checked iterates the incoming map)
          FxOrder agg; bool first = true;
          for (int mi : hit_members) {
            const TagHeader& h = L.tags[(size_t)mem[(size_t)mi]];
            FxOrder tagmap(inferred ? h.codes.size() : 1);
            for (uint32_t c : h.codes) tagmap.insert(c, local_of(c));
            if (first) { agg = tagmap; first = false; }
            else { auto cs = tagmap.codes(); auto os = tagmap.order(); for (size_t i = 0; i < cs.size(); i++) agg.insert(cs[i], os[i]); }
          }
          std::vector<int> pre = agg.order(); std::vector<uint32_t> prec = agg.codes();
          std::vector<int> post = pre;
          if (collapse) { FxOrder nm; for (size_t i = 0; i < pre.size(); i++) if (prec[i] != cc.collapse_code) nm.insert(prec[i], pre[i]);
            post = nm.order(); }
          uint32_t v = (uint32_t)pre.size() | ((uint32_t)post.size() << 3);
          for (size_t i = 0; i < pre.size(); i++) v |= (uint32_t)pre[i] << (8 + 2 * i);
          for (size_t i = 0; i < post.size(); i++) v |= (uint32_t)post[i] << (16 + 2 * i);
          G.pat[pat] = v;
        };
        for (int pat = 1; pat < (1 << mem.size()); pat++) { std::vector<int> hm;
          for (size_t mi = 0; mi < mem.size(); mi++) if (pat & (1 << mi)) hm.push_back((int)mi);
          fill(pat, hm, false); }
        if (implicit) { std::vector<int> hm; for (size_t mi = 0; mi < mem.size(); mi++) if (implicit & (1u << mi)) hm.push_back((int)mi);
          fill(MKP_PAT_INFERRED, hm, true); }
      }
      dev.push_back(D);
    }
  }
};

// ------------------------------------------------------------------------------------------
// std::vector whose resize() leaves new elements uninitialised (the caller fills them; pages are first touched by the writer)
// Buffers of 8 MiB and more are mapped directly and advised to use huge pages: a shard's arrays are hundreds of MB that all cores touch
// for the first time at once, and with 4 KiB pages the first shard of a run spent more time in page faults than in packing.
template <class T> struct DefaultInitAlloc : std::allocator<T> {
  template <class U> struct rebind { using other = DefaultInitAlloc<U>; };
  static constexpr size_t kMapFrom = (size_t)8 << 20, kHuge = (size_t)2 << 20;
  static size_t mapped_len(size_t n) { return (n * sizeof(T) + kHuge - 1) & ~(kHuge - 1); }
  T* allocate(size_t n) {
    if (n * sizeof(T) < kMapFrom) return std::allocator<T>::allocate(n);
    void* m = mmap(nullptr, mapped_len(n), PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    if (m == MAP_FAILED) throw std::bad_alloc();
    madvise(m, mapped_len(n), MADV_HUGEPAGE);
    return (T*)m;
  }
  void deallocate(T* p, size_t n) {
    if (n * sizeof(T) < kMapFrom) std::allocator<T>::deallocate(p, n); else munmap((void*)p, mapped_len(n));
  }
  template <class U, class... A> void construct(U* p, A&&... a) {
    if constexpr (sizeof...(A) == 0) ::new ((void*)p) U; else ::new ((void*)p) U(std::forward<A>(a)...);
  }
};
template <class T> using PodVec = std::vector<T, DefaultInitAlloc<T>>;

struct ShardHost {
  int32_t tid = -1; int32_t win_start = 0, win_end = 0;
  PodVec<MkpReadHdr> hdr; PodVec<uint32_t> cigar; PodVec<uint32_t> chunk_pfx; PodVec<uint8_t> seq; PodVec<MkpTagRef> tagref;
  PodVec<uint32_t> ranks; PodVec<uint8_t> ml;
  uint64_t n_events_cap = 0, n_calls = 0;
  PodVec<uint64_t> name_hash;  // for duplicate-qname detection (read cache is keyed by name, read_cache.rs:28-35)
  // reference spans of records htslib's pileup buffers but the path drops (supplementary): max-depth guard only
  std::vector<std::pair<int32_t, int32_t>> extra_spans;
  // Device ingest (mkp_ingest.hip): cigar / chunk_pfx / seq / ranks / ml were written in HBM and never existed on the host; hdr, tagref
  // (MKP_MAX_TAGS entries per read, pad = "same delta list as the tag before") and name_hash are the digest the planner works from.
  // per read: the probability-sum test of a two-tag read (what make_resident computes from S.ml otherwise)
  bool dev_packed = false; std::vector<uint8_t> dev_sum2;
  uint64_t dev_n_ranks = 0, dev_n_ml = 0;
  std::vector<uint64_t> dev_name_hash2; std::vector<uint32_t> dev_win_idx;   // second name hash; place in the window (file order)
  // records of the window only the threshold sampler takes (QC-fail, CIGAR-less): packed behind the kept ones in the same HBM arrays; their
  // tag table entries follow the kept reads' in `tagref` (tag_off = MKP_MAX_TAGS * (hdr.size() + k))
  PodVec<MkpReadHdr> so_hdr; std::vector<uint64_t> so_name_hash, so_name_hash2; std::vector<uint32_t> so_win_idx;
  // append shard pieces packed independently (parallel packing), in order: offsets are rebased, layout ids remapped.  Sizes
  // are fixed first, then every piece is copied into place by its own thread.
  void append_all(const std::vector<ShardHost>& ps, const std::vector<std::vector<uint16_t>>& layout_maps) {
    struct Base { size_t hdr, cigar, chunk, seq, tag, rank, ml, name; uint64_t ev; };
    std::vector<Base> b(ps.size() + 1);
    b[0] = {hdr.size(), cigar.size(), chunk_pfx.size(), seq.size(), tagref.size(), ranks.size(), ml.size(), name_hash.size(), n_events_cap};
    uint64_t calls = n_calls;
    for (size_t i = 0; i < ps.size(); i++) {
      const ShardHost& o = ps[i];
      b[i + 1] = {b[i].hdr + o.hdr.size(), b[i].cigar + o.cigar.size(), b[i].chunk + o.chunk_pfx.size(), b[i].seq + o.seq.size(),
          b[i].tag + o.tagref.size(),
                  b[i].rank + o.ranks.size(), b[i].ml + o.ml.size(), b[i].name + o.name_hash.size(), b[i].ev + o.n_events_cap};
      calls += o.n_calls;
    }
    const Base& e = b[ps.size()];
    if (e.seq > 0xfffffff0ull || e.cigar > 0xfffffff0ull || e.rank > 0xfffffff0ull || e.ml > 0xfffffff0ull) throw Error(MKP_E_UNSUPPORTED,
        "shard exceeds 4 GiB of packed bases; use smaller shards");
    if (e.ev > 0xfffffff0ull) throw Error(MKP_E_UNSUPPORTED, "shard exceeds 4 Gi call events; use smaller shards");
    hdr.resize(e.hdr); cigar.resize(e.cigar); chunk_pfx.resize(e.chunk); seq.resize(e.seq); tagref.resize(e.tag); ranks.resize(e.rank);
      ml.resize(e.ml); name_hash.resize(e.name);
    auto place = [&](size_t i) {
      const ShardHost& o = ps[i]; const Base& at = b[i]; const std::vector<uint16_t>& lm = layout_maps[i];
      auto cp = [](auto& dst, size_t off, const auto& src) { if (!src.empty()) memcpy(dst.data() + off, src.data(), src.size() * sizeof(src[0])); };
      cp(cigar, at.cigar, o.cigar); cp(chunk_pfx, at.chunk, o.chunk_pfx); cp(seq, at.seq, o.seq); cp(ranks, at.rank, o.ranks); cp(ml, at.ml, o.ml);
        cp(name_hash, at.name, o.name_hash);
      for (size_t k = 0; k < o.hdr.size(); k++) {
        MkpReadHdr h = o.hdr[k];
        h.cigar_off += (uint32_t)at.cigar; h.chunk_off += (uint32_t)(at.chunk / 2); h.seq_off += (uint32_t)at.seq; h.tag_off += (uint32_t)at.tag;
          h.event_off += (uint32_t)at.ev;
        if (h.n_tags) h.layout = lm[h.layout];
        hdr[at.hdr + k] = h;
      }
      for (size_t k = 0; k < o.tagref.size(); k++) { MkpTagRef t = o.tagref[k]; t.rank_off += (uint32_t)at.rank; t.ml_off += (uint32_t)at.ml;
        tagref[at.tag + k] = t; }
    };
    HostPool::get().parallel(ps.size(), place);
    n_events_cap = e.ev; n_calls = calls;
  }
  void clear() { hdr.clear(); cigar.clear(); chunk_pfx.clear(); seq.clear(); tagref.clear(); ranks.clear(); ml.clear(); n_events_cap = 0; n_calls = 0;
    name_hash.clear(); extra_spans.clear();
               dev_packed = false; dev_sum2.clear(); dev_n_ranks = dev_n_ml = 0; dev_name_hash2.clear(); dev_win_idx.clear(); so_hdr.clear();
                 so_name_hash.clear(); so_name_hash2.clear(); so_win_idx.clear(); }
};

class Packer {
 public:
  std::vector<LayoutHost> layouts;
  std::vector<std::string> layout_keys;   // key of layouts[i]
  std::unordered_map<std::string, uint16_t> layout_ids;
  // pieces of the all-cores pack (pack_records): kept between shards so that their buffers — as large as a shard, together — are
  // mapped and faulted in once instead of once per shard (fresh mappings cost page faults on 64 threads and TLB shootdowns on unmap)
  std::vector<ShardHost> pieces;

  // intern the layouts of an independently used packer; returns its id -> this packer's id (first-appearance order is kept)
  std::vector<uint16_t> adopt(const Packer& o) {
    std::vector<uint16_t> map(o.layouts.size());
    for (size_t i = 0; i < o.layouts.size(); i++) {
      auto it = layout_ids.find(o.layout_keys[i]);
      if (it == layout_ids.end()) {
        if (layouts.size() >= 65535) throw Error(MKP_E_UNSUPPORTED, "too many distinct MM header structures");
        layouts.push_back(o.layouts[i]); layout_keys.push_back(o.layout_keys[i]);
        it = layout_ids.emplace(o.layout_keys[i], (uint16_t)(layouts.size() - 1)).first;
      }
      map[i] = it->second;
    }
    return map;
  }

  // aux scan (bam_aux_get: first occurrence); returns pointer at the type byte or null
  static const uint8_t* aux_find(const uint8_t* aux, size_t n, char t0, char t1) {
    size_t o = 0;
    while (o + 3 <= n) {
      char ty = (char)aux[o + 2]; size_t v = o + 3, len;
      switch (ty) {
        case 'A': case 'c': case 'C': len = 1; break;
        case 's': case 'S': len = 2; break;
        case 'i': case 'I': case 'f': len = 4; break;
        case 'd': len = 8; break;
        case 'Z': case 'H': { size_t k = v; while (k < n && aux[k]) k++; len = k - v + 1; break; }
        case 'B': { if (v + 5 > n) return nullptr; char st = (char)aux[v]; uint32_t c; memcpy(&c, aux + v + 1, 4); size_t es = (st == 'c'
            || st == 'C') ? 1 : (st == 's' || st == 'S') ? 2 : 4; len = 5 + es * (size_t)c; break; }
        default: return nullptr;
      }
      if (v + len > n) return nullptr;
      if ((char)aux[o] == t0 && (char)aux[o + 1] == t1) return aux + o + 2;
      o = v + len;
    }
    return nullptr;
  }

  // The same answers as aux_find for several tags from ONE walk over the aux block (the MM string of a 10 kb read is a kilobyte that
  // five separate lookups each stepped through byte by byte): out[k] = first occurrence of tags[k] reached before any malformed
  // field, else null — exactly what aux_find returns for each of them.
  static void aux_find_all(const uint8_t* aux, size_t n, const char (*tags)[2], int n_tags, const uint8_t** out) {
    for (int k = 0; k < n_tags; k++) out[k] = nullptr;
    int missing = n_tags; size_t o = 0;
    while (o + 3 <= n && missing) {
      const char ty = (char)aux[o + 2]; const size_t v = o + 3; size_t len;
      switch (ty) {
        case 'A': case 'c': case 'C': len = 1; break;
        case 's': case 'S': len = 2; break;
        case 'i': case 'I': case 'f': len = 4; break;
        case 'd': len = 8; break;
        // (no terminator: runs past the end, malformed below)
        case 'Z': case 'H': { const void* z = v < n ? memchr(aux + v, 0, n - v) : nullptr; len = z ? (size_t)((const uint8_t*)z - (aux + v)) + 1
            : n - v + 1; break; }
        case 'B': { if (v + 5 > n) return; const char st = (char)aux[v]; uint32_t c; memcpy(&c, aux + v + 1, 4); const size_t es = (st == 'c'
            || st == 'C') ? 1 : (st == 's' || st == 'S') ? 2 : 4; len = 5 + es * (size_t)c; break; }
        default: return;
      }
      if (v + len > n) return;
      for (int k = 0; k < n_tags; k++) if (!out[k] && (char)aux[o] == tags[k][0] && (char)aux[o + 1] == tags[k][1]) { out[k] = aux + o + 2; missing--;
        }
      o = v + len;
    }
  }

  // Appends one record.  Records the packer never needs (flag mask of the htslib pileup engine,
  // supplementary, empty SEQ — pileup/mod.rs:783-791 & BAM_DEF_MASK) must be filtered by the caller
  // with `keep()`.
  static bool keep(const mkp_record& r) { return !(r.flag & (4 | 256 | 512 | 1024 | 2048)) && r.l_qseq > 0 && r.n_cigar > 0; }

  void add(const mkp_record& r, ShardHost& S) {
    MkpReadHdr h; memset(&h, 0, sizeof(h));
    if (!r.data
        || r.l_data < 0
        || r.l_qseq < 0 || (uint64_t)r.l_qname + 4ull * r.n_cigar + ((uint64_t)r.l_qseq + 1) / 2 + (uint64_t)r.l_qseq > (uint64_t)r.l_data)
      throw Error(MKP_E_INVALID, "record data shorter than its fields");
    const uint8_t* cg = r.data + r.l_qname;
    const uint8_t* sq = cg + 4 * (size_t)r.n_cigar;
    const uint8_t* aux = sq + ((size_t)r.l_qseq + 1) / 2 + (size_t)r.l_qseq;
    if (aux > r.data + r.l_data) throw Error(MKP_E_INVALID, "record data shorter than its fields");
    size_t aux_n = (size_t)(r.data + r.l_data - aux);
    int64_t reflen = 0, qlen = 0;
    h.cigar_off = (uint32_t)S.cigar.size();
    uint32_t n_cigar = r.n_cigar;
    // unaligned record (sampling only): one soft clip
    if (n_cigar == 0) { S.cigar.push_back(((uint32_t)r.l_qseq << 4) | 4u); n_cigar = 1; qlen = r.l_qseq; }
    h.chunk_off = (uint32_t)(S.chunk_pfx.size() / 2);
    if (r.n_cigar == 0) { S.chunk_pfx.push_back(0); S.chunk_pfx.push_back(0); }
    for (uint32_t k = 0; k < r.n_cigar; k++) { uint32_t w; memcpy(&w, cg + 4 * k, 4); S.cigar.push_back(w); uint32_t op = w & 15;
      if ((k & 63u) == 0) { S.chunk_pfx.push_back((uint32_t)qlen); S.chunk_pfx.push_back((uint32_t)reflen);
        } if (op == 0 || op == 2 || op == 3 || op == 7 || op == 8) reflen += w >> 4;
        if (op == 0 || op == 1 || op == 4 || op == 7 || op == 8) qlen += w >> 4;
        }
    if (qlen != r.l_qseq) throw Error(MKP_E_INVALID, "CIGAR query length does not match SEQ length");
    if (qlen >= (1 << 26) || reflen >= (1 << 26)) throw Error(MKP_E_UNSUPPORTED,
        "a read or its alignment spans 2^26 bases or more (the depth walk packs query offsets in 27 bits)");
    h.ref_start = r.pos; h.ref_end = r.pos + (int32_t)reflen; h.l_seq = (uint32_t)r.l_qseq; h.n_cigar = n_cigar;
    if (S.seq.size() + (size_t)r.l_qseq / 2 + 8 > 0xfffffff0ull || S.cigar.size() > 0xfffffff0ull) throw Error(MKP_E_UNSUPPORTED,
        "shard exceeds 4 GiB of packed bases; use smaller shards");
    h.seq_off = (uint32_t)S.seq.size();
    S.seq.insert(S.seq.end(), sq, sq + ((size_t)r.l_qseq + 1) / 2);
    while (S.seq.size() & 3) S.seq.push_back(0);
    h.flags = (r.flag & 16) ? MKP_RF_REVERSE : 0;
    h.tag_off = (uint32_t)S.tagref.size();
    h.event_off = (uint32_t)S.n_events_cap;
    { uint64_t hh = 1469598103934665603ull; for (int i = 0; i + 1 < r.l_qname; i++) { hh ^= r.data[i]; hh *= 1099511628211ull;
      } S.name_hash.push_back(hh); }
    size_t rank_mark = S.ranks.size(), ml_mark = S.ml.size(), tag_mark = S.tagref.size();
    uint64_t cap = 0;
    if (!tokenise(r, aux, aux_n, S, &h, &cap)) {  // tag error: the read only contributes coverage (read_cache.rs:272-277)
      S.ranks.resize(rank_mark); S.ml.resize(ml_mark); S.tagref.resize(tag_mark);
      h.flags |= MKP_RF_BAD; h.n_tags = 0; cap = 0;
    }
    h.event_cap = (uint32_t)cap;
    S.n_events_cap += cap;
    if (S.n_events_cap > 0xfffffff0ull) throw Error(MKP_E_UNSUPPORTED, "shard exceeds 4 Gi call events; use smaller shards");
    S.hdr.push_back(h);
  }

 private:
  static bool ws(char c) { return c == ' ' || c == '\t' || c == '\n' || c == '\r' || c == '\f' || c == '\v'; }

  bool tokenise(const mkp_record& r, const uint8_t* aux, size_t aux_n, ShardHost& S, MkpReadHdr* h, uint64_t* cap) {
    // get_tag: new style wins, each looked up independently (util.rs:174-188)
    static const char want[5][2] = {{'M', 'M'}, {'M', 'm'}, {'M', 'L'}, {'M', 'l'}, {'M', 'N'}};
    const uint8_t* at[5]; aux_find_all(aux, aux_n, want, 5, at);
    const uint8_t* mm = at[0] ? at[0] : at[1];
    const uint8_t* mlp = at[2] ? at[2] : at[3];
    if (!mm || !mlp) return false;
    if ((char)mm[0] != 'Z') return false;
    if (!((char)mlp[0] == 'B' && (char)mlp[1] == 'C')) return false;
    uint32_t ml_n; memcpy(&ml_n, mlp + 2, 4);
    const uint8_t* mn = at[4];
    if (mn) {
      int64_t v;
      switch ((char)mn[0]) {
        case 'c': v = (int8_t)mn[1]; break; case 'C': v = mn[1]; break;
        case 's': { int16_t x; memcpy(&x, mn + 1, 2); v = x; break; } case 'S': { uint16_t x; memcpy(&x, mn + 1, 2); v = x; break; }
        case 'i': { int32_t x; memcpy(&x, mn + 1, 4); v = x; break; } case 'I': { uint32_t x; memcpy(&x, mn + 1, 4); v = x; break; }
        default: return false;
      }
      if ((uint64_t)v != (uint64_t)(uint32_t)r.l_qseq) return false;  // check_mn_tag_correct (mod_bam.rs:1431-1449)
    // NonPrimaryMissingMn: a secondary / supplementary / duplicate record needs MN (1444-1446)
    } else if (r.flag & (256 | 1024 | 2048)) return false;
    const char* s = (const char*)mm + 1;
    std::string key; std::vector<TagHeader> hdrs; std::vector<MkpTagRef> refs;
    uint32_t ml_base = (uint32_t)S.ml.size(); uint64_t pointer = 0; uint64_t calls = 0; bool implicit_strand[2] = {false, false};
    while (*s) {
      const char* e = s; while (*e && *e != ';') e++;
      if (e > s) {
        // ---- header (MmTagInfo::parse, mod_bam.rs:909-983)
        const char* p = s; const char* he = s; while (he < e && *he != ',') he++;
        TagHeader th;
        if (p >= he) return false;
        switch (*p) { case 'A': th.fb = 0; break; case 'C': th.fb = 1; break; case 'G': th.fb = 2; break; case 'T': case 'U': th.fb = 3; break;
          case 'N': th.fb = 4; break; default: return false; }
        p++; if (p >= he) return false;
        if (*p == '+') th.neg = false; else if (*p == '-') th.neg = true; else return false;
        p++; th.mode = 2; bool chebi = false; size_t offset = 2;
        if (p < he && *p >= '0' && *p <= '9') { uint64_t v = 0; while (p < he && *p >= '0' && *p <= '9') { v = v * 10 + (uint64_t)(*p - '0');
            if (v > 0x7fffffffull) return false;
            p++; offset++; } th.codes.push_back(0x80000000u | (uint32_t)v); chebi = true; }
        for (; p < he; p++) {
          if (*p == '?' || *p == '.') { th.mode = *p == '?' ? 0 : 1; offset++; }
          else if (*p >= '0' && *p <= '9') return false;
          else { if (chebi) return false; if ((unsigned char)*p >= 0x80) throw Error(MKP_E_UNSUPPORTED, "non-ASCII mod code");
            th.codes.push_back((uint32_t)(unsigned char)*p); offset++; }
        }
        if (th.codes.size() > MKP_KMAX) throw Error(MKP_E_UNSUPPORTED, "more than 4 mod codes in one MM tag");
        // ---- delta list -> cumulative ranks (to_positions_specific / to_positions, mod_bam.rs:697-767)
        MkpTagRef tr; tr.rank_off = (uint32_t)S.ranks.size(); tr.n = 0; tr.ml_off = 0; tr.pad = 0;
        if (offset + 1 <= (size_t)(e - s)) {
          const char* d = s + offset + 1; bool first = true; uint64_t acc = 0;
          for (;;) {
            const char* save = d;
            if (!first) { if (d >= e || *d != ',') break; d++; }
            while (d < e && ws(*d)) d++;
            if (!(d < e && *d >= '0' && *d <= '9')) { if (first) return false; d = save; break; }
            uint64_t v = 0; while (d < e && *d >= '0' && *d <= '9') { v = v * 10 + (uint64_t)(*d - '0'); if (v > 0xffffffffull) return false; d++; }
            while (d < e && ws(*d)) d++;
            acc = first ? v : acc + v + 1;  // Σ(d+1)-1
            if (acc >= (th.fb == 4 ? (uint64_t)(uint32_t)r.l_qseq : 0xffffffffull)) { if (th.fb == 4 || acc >= 0xffffffffull) return false; }
            S.ranks.push_back((uint32_t)acc); tr.n++; first = false;
          }
        }
        if (th.codes.empty() && tr.n) { /* stride 0: the reference's chunks(0) would panic */ return false; }
        uint64_t need = pointer + (uint64_t)tr.n * th.codes.size();
        if (need > ml_n) return false;  // "ML array too short" (mod_bam.rs:1222-1228)
        tr.ml_off = ml_base + (uint32_t)pointer; pointer = need; calls += tr.n;
        if (th.mode != 0 && th.fb != 4) implicit_strand[th.neg ? 1 : 0] = true;
        key.push_back("ACGTN"[th.fb]); key.push_back(th.neg ? '-' : '+'); key.push_back((char)('0' + th.mode));
        for (uint32_t c : th.codes) { key.append(std::to_string(c)); key.push_back('/'); }
        key.push_back(';');
        hdrs.push_back(std::move(th)); refs.push_back(tr);
      }
      s = *e ? e + 1 : e;
    }
    if (hdrs.empty()) return false;  // no tags -> ModBaseInfo::is_empty -> NoModifiedBaseInformation
    if (hdrs.size() > MKP_MAX_TAGS) throw Error(MKP_E_UNSUPPORTED, "more than 8 MM tags in one read");
    auto it = layout_ids.find(key);
    if (it == layout_ids.end()) {
      if (layouts.size() >= 65535) throw Error(MKP_E_UNSUPPORTED, "too many distinct MM header structures");
      LayoutHost L; L.tags = hdrs; layouts.push_back(std::move(L)); layout_keys.push_back(key);
      it = layout_ids.emplace(key, (uint16_t)(layouts.size() - 1)).first;
    }
    h->layout = it->second; h->n_tags = (uint16_t)hdrs.size();
    for (auto& tr : refs) S.tagref.push_back(tr);
    S.ml.insert(S.ml.end(), mlp + 6, mlp + 6 + pointer);
    S.n_calls += calls;
    *cap = calls + (uint64_t)(uint32_t)r.l_qseq * ((implicit_strand[0] ? 1 : 0) + (implicit_strand[1] ? 1 : 0));
    return true;
  }
};

// Pack `recs` (those `keep` accepts) behind `dst`.  MM tokenising dominates packing: above a few thousand records contiguous
// record ranges are packed independently by all host cores and appended in order (same layout ids and offsets as one
// sequential pass).
template <class Keep> void pack_records(Packer& packer, ShardHost& dst, const mkp_record* recs, uint32_t n, Keep keep, uint32_t min_parallel = 1024) {
  unsigned hw = HostPool::host_cpus();
  if (const char* e = getenv("MKP_PACK_PIECES")) hw = std::max(1u, std::min(4096u, (unsigned)strtoul(e, nullptr, 10)));   // experiments
  const unsigned n_thr = n >= min_parallel ? std::max(1u, std::min(hw, n)) : 1;
  if (n_thr == 1) { for (uint32_t i = 0; i < n; i++) if (keep(recs[i])) packer.add(recs[i], dst); return; }
  std::vector<Packer> pk(n_thr); std::vector<std::unique_ptr<Error>> errs(n_thr);
  if (packer.pieces.size() != n_thr) { packer.pieces.clear(); packer.pieces.resize(n_thr); }
  std::vector<ShardHost>& sh = packer.pieces;
  HostPool::get().parallel(n_thr, [&](size_t t) {
    sh[t].clear();
    const uint32_t lo = (uint32_t)((uint64_t)n * t / n_thr), hi = (uint32_t)((uint64_t)n * (t + 1) / n_thr);
    sh[t].tid = dst.tid; sh[t].win_start = dst.win_start; sh[t].win_end = dst.win_end;
    {   // room for the piece's bases and CIGARs up front (known from the record cores): no regrowth copies while packing
      size_t seq_b = 0, cig = 0;
      for (uint32_t i = lo; i < hi; i++) { seq_b += (((size_t)std::max(recs[i].l_qseq, 0) + 1) / 2 + 3) & ~(size_t)3; cig += recs[i].n_cigar; }
      sh[t].seq.reserve(seq_b + 64); sh[t].cigar.reserve(cig + 64); sh[t].hdr.reserve(hi - lo); sh[t].name_hash.reserve(hi - lo);
    }
    try { for (uint32_t i = lo; i < hi; i++) if (keep(recs[i])) pk[t].add(recs[i], sh[t]); }
    catch (const Error& e) { errs[t].reset(new Error(e)); }
    catch (const std::exception& e) { errs[t].reset(new Error(MKP_E_INVALID, e.what())); }
  });
  std::vector<std::vector<uint16_t>> maps(n_thr);
  for (unsigned t = 0; t < n_thr; t++) { if (errs[t]) throw *errs[t]; maps[t] = packer.adopt(pk[t]); }
  dst.append_all(sh, maps);
}

}  // namespace mkp

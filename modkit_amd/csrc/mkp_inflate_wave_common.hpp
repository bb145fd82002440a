// Pieces shared by the wave-per-block inflate kernels (mkp_inflate_wave.hip, mkp_inflate_wave2.hip): the LDS ordering macro, the
// code-table builder, RFC 1951's length / distance base tables as arithmetic, the ring -> global flush.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "mkp_inflate_tok.hpp"

namespace {
constexpr uint32_t RING = 32768u, LIT_BITS = 11u, DIST_BITS = 9u;

// order this wave's LDS traffic only: a fence without the address space also drains the global stores of the output bytes (vmcnt(0)),
// ~1 us at every match — that made the first version of this kernel 2.4x SLOWER than one thread per block
#define LDS_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local"); } while (0)

__device__ __forceinline__ uint32_t sgpr(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }

// direct table + canonical lists from lens[0, n): all lanes.  Returns 0 complete, > 0 incomplete, < 0 over-subscribed.
__device__ __forceinline__ int build(const uint8_t* lens, int n, uint16_t* tab, uint32_t tab_bits, uint16_t* count, uint16_t* syms, int lane) {
  if (lane < 16) count[lane] = 0;
  for (uint32_t i = (uint32_t)lane; i < (1u << tab_bits); i += 64u) tab[i] = 0;
  LDS_SYNC();
  for (int s = lane; s < n; s += 64) { const uint32_t l = lens[s]; if (l) atomicAdd(reinterpret_cast<uint32_t*>(count) + (l >> 1), (l & 1u) ? 0x10000u : 1u); }   // two u16 counters per dword
  LDS_SYNC();
  uint32_t next_code[16], offs[16]; int left = 1; uint32_t code = 0, off = 0; uint32_t used = 0;
  next_code[0] = 0; offs[0] = 0;
  for (int l = 1; l <= 15; l++) {
    const uint32_t c = sgpr(count[l]);
    left = (left << 1) - (int)c; code = (code + (l > 1 ? sgpr(count[l - 1]) : 0u)) << 1; next_code[l] = code; offs[l] = off; off += c; used += c;
  }
  if (left < 0) return left;
  if (used == 0) return 0;
  // symbols in order: the code of a symbol is next_code[len]++ (uniform walk); its table slots are filled by all lanes
  for (int s = 0; s < n; s++) {
    const uint32_t l = sgpr(lens[s]); if (!l) continue;
    uint32_t c = 0, o = 0;
#pragma unroll
    for (int k = 1; k <= 15; k++) if ((uint32_t)k == l) { c = next_code[k]++; o = offs[k]++; }   // (register arrays: constant indices only)
    if (lane == 0) syms[o] = (uint16_t)s;
    if (l <= tab_bits) {
      const uint32_t rev = __builtin_bitreverse32(c) >> (32u - l);
      const uint16_t ent = (uint16_t)(l | ((uint32_t)s << 4));
      for (uint32_t k = rev + ((uint32_t)lane << l); k < (1u << tab_bits); k += 64u << l) tab[k] = ent;
    }
  }
  LDS_SYNC();
  return left;
}

__device__ __forceinline__ uint32_t cl_order(int i) {
  const unsigned long long lo = 16ull | (17ull << 5) | (18ull << 10) | (0ull << 15) | (8ull << 20) | (7ull << 25) | (9ull << 30) | (6ull << 35) | (10ull << 40) | (5ull << 45) | (11ull << 50) | (4ull << 55);
  const unsigned long long hi = 12ull | (3ull << 5) | (13ull << 10) | (2ull << 15) | (14ull << 20) | (1ull << 25) | (15ull << 30);
  return (uint32_t)((i < 12 ? lo >> (5 * i) : hi >> (5 * (i - 12))) & 31ull);
}
// ring -> global, output bytes [from, to): 16 bytes per lane and step (from is a multiple of 16 unless it follows a stored block)
__device__ __forceinline__ void flush(const uint8_t* ring, uint8_t* __restrict__ o, uint32_t from, uint32_t to, int lane) {
  uint32_t a = from;
  if (a & 15u) { const uint32_t head = min(to, (a + 15u) & ~15u); for (uint32_t k = a + (uint32_t)lane; k < head; k += 64u) o[k] = ring[k & (RING - 1u)]; a = head; }
  const uint32_t units = (to - a) >> 4;
  for (uint32_t u = (uint32_t)lane; u < units; u += 64u) { const uint32_t at = a + 16u * u; uint4 v = *reinterpret_cast<const uint4*>(ring + (at & (RING - 1u))); __builtin_memcpy(o + at, &v, 16); }
  for (uint32_t k = a + 16u * units + (uint32_t)lane; k < to; k += 64u) o[k] = ring[k & (RING - 1u)];
}
}  // namespace

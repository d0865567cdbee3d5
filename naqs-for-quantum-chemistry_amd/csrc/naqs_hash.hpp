// naqs_hash.hpp — device-side hash table of the sample keys, shared by the E_loc kernels (naqs_hip.hip) and by
// the fused log-psi + E_loc entry point (naqs_logpsi.hip inserts the keys from its amplitude kernel).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace naqs {

// Open-addressing hash table of the sample keys, rebuilt every call WITHOUT clearing it: each slot
// carries the 8-bit epoch of the call that wrote it next to the 24-bit sample index, and a slot whose
// epoch is not the current one counts as empty.  (The table is zeroed when it is allocated and when
// the epoch wraps, every 255 calls.)  32-bit keys: one 8-byte word  key << 32 | epoch << 24 | index.
constexpr uint32_t IDX_MASK = 0x00FFFFFFu;
template <typename KT> struct Slot;
template <> struct Slot<uint32_t> { unsigned long long kv; };
template <> struct Slot<uint64_t> { unsigned long long key; uint32_t val; uint32_t pad; };   // val = epoch << 24 | index

__device__ __forceinline__ int popc(uint32_t x) { return __popc(x); }
__device__ __forceinline__ int popc(uint64_t x) { return __popcll(x); }

__device__ __forceinline__ uint32_t hash_key(uint32_t k, int bits) { return (k * 0x9E3779B1u) >> (32 - bits); }
__device__ __forceinline__ uint32_t hash_key(uint64_t k, int bits) {
    return (uint32_t)((k * 0x9E3779B97F4A7C15ull) >> (64 - bits));
}

__device__ __forceinline__ void hash_insert(Slot<uint32_t> *tab, int bits, uint32_t tag, uint32_t key, uint32_t idx) {
    const uint32_t mask = (1u << bits) - 1u;
    uint32_t h = hash_key(key, bits);
    const unsigned long long want = ((unsigned long long)key << 32) | tag | idx;
    unsigned long long old = tab[h].kv;
    for (;;) {
        if (((uint32_t)old & ~IDX_MASK) == tag) {          // taken in this epoch -> next slot
            h = (h + 1) & mask;
            old = tab[h].kv;
            continue;
        }
        const unsigned long long prev = atomicCAS(&tab[h].kv, old, want);
        if (prev == old) return;
        old = prev;                                         // lost the race for this slot: look again
    }
}
__device__ __forceinline__ void hash_insert(Slot<uint64_t> *tab, int bits, uint32_t tag, uint64_t key, uint32_t idx) {
    const uint32_t mask = (1u << bits) - 1u;
    uint32_t h = hash_key(key, bits);
    uint32_t old = tab[h].val;
    for (;;) {
        if ((old & ~IDX_MASK) == tag) {
            h = (h + 1) & mask;
            old = tab[h].val;
            continue;
        }
        const uint32_t prev = atomicCAS(&tab[h].val, old, tag | idx);
        if (prev == old) { tab[h].key = (unsigned long long)key; return; }   // claimed: readers run in a later kernel
        old = prev;
    }
}

// first probe (the load that matters for latency) and its resolution, split so that callers can put
// several probes in flight before looking at any of them
__device__ __forceinline__ unsigned long long probe_load(const Slot<uint32_t> *__restrict__ tab, uint32_t h) { return tab[h].kv; }
__device__ __forceinline__ int probe_resolve(const Slot<uint32_t> *__restrict__ tab, int bits, uint32_t tag, uint32_t key,
                                             uint32_t h, unsigned long long kv) {
    const uint32_t mask = (1u << bits) - 1u;
    for (;;) {
        if (((uint32_t)kv & ~IDX_MASK) != tag) return -1;
        if ((uint32_t)(kv >> 32) == key) return (int)((uint32_t)kv & IDX_MASK);
        h = (h + 1) & mask;
        kv = tab[h].kv;
    }
}
struct Slot64Val { unsigned long long key; uint32_t val; };
__device__ __forceinline__ Slot64Val probe_load(const Slot<uint64_t> *__restrict__ tab, uint32_t h) {
    const Slot<uint64_t> s = tab[h];
    return Slot64Val{s.key, s.val};
}
__device__ __forceinline__ int probe_resolve(const Slot<uint64_t> *__restrict__ tab, int bits, uint32_t tag, uint64_t key,
                                             uint32_t h, Slot64Val s) {
    const uint32_t mask = (1u << bits) - 1u;
    for (;;) {
        if ((s.val & ~IDX_MASK) != tag) return -1;
        if (s.key == key) return (int)(s.val & IDX_MASK);
        h = (h + 1) & mask;
        s = probe_load(tab, h);
    }
}
template <typename KT>
__device__ __forceinline__ int hash_find(const Slot<KT> *__restrict__ tab, int bits, uint32_t tag, KT key) {
    const uint32_t h = hash_key(key, bits);
    return probe_resolve(tab, bits, tag, key, h, probe_load(tab, h));
}


// Optional Bloom filter over the sample keys (2^19 bits = 64 KiB): the E_loc kernel keeps
// a copy in LDS and only sends candidates whose bit is set to the hash table in L2.  Pays off when most
// physical candidates are NOT in the sample set (large Hilbert spaces: Li2O, M = 5*10^4 of 4*10^7 states).
constexpr int BLOOM_LOG2_BITS = 19;
constexpr int BLOOM_WORDS = 1 << (BLOOM_LOG2_BITS - 5);
// Blocked filter: one multiplicative hash picks a 32-bit word (top 14 bits) and three bit positions inside it (the next
// 15 bits), so a membership test is ONE LDS read and an and/compare.  History: a single bit per key let 9.5 % of the
// absent candidates through to the hash table in L2 (5*10^4 keys) and those probes were the kernel (452 us at Li2O);
// three independent hashes cut that to 1.5 % but their address arithmetic made the (VALU-bound) candidate loop longer
// (355 us); the blocked form keeps ~2 % false positives at a third of the instructions.
__device__ __forceinline__ uint32_t bloom_hash(uint32_t k) { return k * 0x85EBCA6Bu; }
__device__ __forceinline__ uint32_t bloom_hash(uint64_t k) { return (uint32_t)((k * 0xC2B2AE3D27D4EB4Full) >> 32); }
__device__ __forceinline__ uint32_t bloom_word(uint32_t h) { return h >> (32 - (BLOOM_LOG2_BITS - 5)); }
__device__ __forceinline__ uint32_t bloom_mask(uint32_t h) {
    return (1u << ((h >> 13) & 31u)) | (1u << ((h >> 8) & 31u)) | (1u << ((h >> 3) & 31u));
}
template <typename KT>
__device__ __forceinline__ void bloom_insert(uint32_t *bloom, KT key) {
    const uint32_t h = bloom_hash(key);
    atomicOr(&bloom[bloom_word(h)], bloom_mask(h));
}
template <typename KT>
__device__ __forceinline__ bool bloom_test(const uint32_t *bloom, KT key) {
    const uint32_t h = bloom_hash(key), m = bloom_mask(h);
    return (bloom[bloom_word(h)] & m) == m;
}

// what a producer kernel outside naqs_hip.hip needs to fill the E_loc scratch of a handle for one call
struct ElocFeed {
    uint32_t *bloom;      // BLOOM_WORDS words, zeroed for this call, or nullptr
    void *tab;            // Slot<uint32_t>* or Slot<uint64_t>*
    void *keys_narrow;    // uint32_t* or uint64_t* [M]
    double2 *psi;         // [M] (Re, Im) f64
    int32_t bits;
    uint32_t tag;
    int32_t key_bits;     // 32 | 64
};

template <typename KT>
__device__ __forceinline__ void feed_key(const ElocFeed &f, int64_t i, uint64_t key) {
    reinterpret_cast<KT *>(f.keys_narrow)[i] = (KT)key;
    hash_insert(reinterpret_cast<Slot<KT> *>(f.tab), f.bits, f.tag, (KT)key, (uint32_t)i);
    if (f.bloom != nullptr) {
        bloom_insert<KT>(f.bloom, (KT)key);
    }
}
__device__ __forceinline__ void feed_psi(const ElocFeed &f, int64_t i, float log_amp, float phase) {
    const double amp = exp((double)log_amp);
    double s, c;
    sincos((double)phase, &s, &c);
    f.psi[i] = make_double2(amp * c, amp * s);
}

}  // namespace naqs

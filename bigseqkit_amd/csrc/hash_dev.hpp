// ============================================================================
// hash_dev.hpp -- the two 64-bit keys of a subject (device code only).
//
//   k1 = XXH64(subject, seed 0) == int64(xxhash.Sum64(subject)), the reference's grouping key
//        (/root/reference/bigseqkit-lib/rmdup.go:67-84);
//   k2 = a second, independent 64-bit hash of the same bytes.  The reference tells the subjects of one XXH64 group
//        apart by comparing the strings (rmdup.go:150-199, a map keyed by the subject); here two records are the
//        same subject iff k1 AND k2 agree (128 bits), so that no record text is read a second time.  k2 is cheap on
//        purpose and laid out like XXH64's stripes -- the 8-byte word at offset 8 j goes to chain j & 3 -- so the four
//        lanes that run the XXH64 accumulators of a subject feed it from the words they already hold:
//            b[k] = Q[(k + 1) & 3];   for every whole 8-byte word w_j:  b[j & 3] = rotl64((b[j & 3] ^ w_j) * Q[j & 3], 31)
//            rest = the last len & 7 bytes, little endian, zero padded
//            t = b0 ^ rotl(b1, 16) ^ rotl(b2, 32) ^ rotl(b3, 48);  t = (t ^ rest) * QF1;  t ^= t >> 32;
//            t = (t + len) * QF2;  t ^= t >> 29;  t *= QF3;  t ^= t >> 32
//        (tests/test_rmdup_keys_gpu.py restates it in Python and holds the device keys to it.)
// ============================================================================
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace bsk {
namespace hashdev {

constexpr uint64_t P1 = 11400714785074694791ull, P2 = 14029467366897019727ull, P3 = 1609587929392839161ull,
                   P4 = 9650029242287828579ull, P5 = 2870177450012600261ull;
constexpr uint64_t Q0 = 0x9E3779B97F4A7C15ull, Q1 = 0xC2B2AE3D27D4EB4Full, Q2 = 0x165667B19E3779F9ull, Q3 = 0xD6E8FEB86659FD93ull;
constexpr uint64_t QF1 = 0x9FB21C651E98DF25ull, QF2 = 0xFF51AFD7ED558CCDull, QF3 = 0xC4CEB9FE1A85EC53ull;

__device__ __forceinline__ uint64_t rotl64(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }
__device__ __forceinline__ uint64_t xround(uint64_t acc, uint64_t in) { return rotl64(acc + in * P2, 31) * P1; }
__device__ __forceinline__ uint64_t xmerge(uint64_t acc, uint64_t v) { return (acc ^ xround(0, v)) * P1 + P4; }
__device__ __forceinline__ uint64_t xavalanche(uint64_t h) {
    h ^= h >> 33; h *= P2; h ^= h >> 29; h *= P3; h ^= h >> 32;
    return h;
}

__device__ __forceinline__ uint64_t k2_q(uint32_t k) { return k == 0 ? Q0 : (k == 1 ? Q1 : (k == 2 ? Q2 : Q3)); }
__device__ __forceinline__ uint64_t k2_init(uint32_t k) { return k2_q((k + 1u) & 3u); }
__device__ __forceinline__ uint64_t k2_step(uint64_t b, uint64_t w, uint64_t q) { return rotl64((b ^ w) * q, 31); }
__device__ __forceinline__ uint64_t k2_finish(uint64_t b0, uint64_t b1, uint64_t b2, uint64_t b3, uint64_t rest, uint64_t len) {
    uint64_t t = b0 ^ rotl64(b1, 16) ^ rotl64(b2, 32) ^ rotl64(b3, 48);
    t = (t ^ rest) * QF1;
    t ^= t >> 32;
    t = (t + len) * QF2;
    t ^= t >> 29;
    t *= QF3;
    t ^= t >> 32;
    return t;
}

// ---- the GROUPING key of the byte-verifying `rmdup -s` (round 5) ------------------------------------------------------
// When the bytes of every duplicate are compared with its survivor's (ops_host_rmdup.cpp, the default), the key only
// GROUPS: its value never reaches the output, so it need not be the reference's XXH64 -- whose four serial accumulator
// chains, merge, tail and avalanche were 40 % of the vector instructions of k_rmdup_stream.  This one has no
// chain at all: the subject is cut into 16-byte chunks (the last one zero padded), chunk c is four dwords w0..w3, and with
// two independent key streams K, K' (eight dwords per chunk position, splitmix64 of the position)
//      a  += (w0 + K0) * (w1 + K1) + (w2 + K2) * (w3 + K3)         (32 x 32 -> 64 bit products: one v_mad_u64_u32 each)
//      a' += (w0 + K0') * (w1 + K1') + (w2 + K2') * (w3 + K3')
// -- the NH family of UMAC (Black, Halevi, Krawczyk, Krovetz, Rogaway 1999): two subjects of equal length collide in one
// 64-bit sum with probability <= 2^-32 over the keys, in both with 2^-64 -- summed over the chunks in ANY order, so the
// four lanes of a quad take chunks c = k, k + 4, ... and add up with two DPP steps.  Positions repeat every 64 chunks
// (1 KiB); a lane that passes its 16th, 32nd, ... chunk stirs its sums (rotate, multiply) so that chunks 1 KiB apart do
// not commute.  The two sums, the length and a multiply-xorshift finish give the key (its LOW bits choose the radix
// bucket).  A collision costs a second round with XXH64 + k2 (the path that exists and is tested), never a wrong answer.
// tests/test_rmdup_keys_gpu.py restates it in Python.
constexpr uint32_t GKEY_POS = 64;                 // chunk positions with keys of their own
constexpr uint32_t GKEY_BYTES = GKEY_POS * 32;    // 2 KiB of LDS per block
__host__ __device__ __forceinline__ uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
// dword j (0..7) of chunk position p: K0..K3 then K0'..K3'
__host__ __device__ __forceinline__ uint32_t gkey_word(uint32_t p, uint32_t j) {
    const uint64_t v = splitmix64(0x6b73625f67726f75ull + (uint64_t)(p * 4u + (j >> 1)));
    return (j & 1u) ? (uint32_t)(v >> 32) : (uint32_t)v;
}
__device__ __forceinline__ uint64_t gkey_stir1(uint64_t a) { return rotl64(a, 29) * P1; }
__device__ __forceinline__ uint64_t gkey_stir2(uint64_t a) { return rotl64(a, 31) * P2; }
__device__ __forceinline__ uint64_t gkey_finish(uint64_t a1, uint64_t a2, uint64_t len) {
    uint64_t h = (a1 + len) * P1 + rotl64(a2, 32) * P2;
    h ^= h >> 32;
    h *= P3;
    h ^= h >> 29;
    return h;
}

// lower8 on four bytes at once
__device__ __forceinline__ uint32_t fold4(uint32_t x) {
    const uint32_t ge_a = (x & 0x7F7F7F7Fu) + 0x3F3F3F3Fu, ge_z1 = (x & 0x7F7F7F7Fu) + 0x25252525u;
    return x | ((ge_a & ~ge_z1 & ~x & 0x80808080u) >> 2);
}

}  // namespace hashdev
}  // namespace bsk

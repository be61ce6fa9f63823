// Host-side helpers of the GPU-resident feeder (no device code).
// Reference: the Sampler re-shuffles its index list with Python's `random.shuffle` at every epoch and repack (util.py:152-157,
// 226-235); every later draw of the run (validation split, exemplar selection, herding ties) continues that ONE Mersenne-Twister
// stream, so a feeder that wants the reference's batches has to consume it draw for draw.  On the shipped datasets the list holds
// 10^5 entries and the pure-Python loop costs 2-11 ms per epoch -- 10 % of an end-to-end run once the train step is 0.4 ms.  This
// is the same shuffle on an int64 array, given the generator's state: CPython's algorithm (Lib/random.py: shuffle ->
// _randbelow_with_getrandbits -> getrandbits(k) = genrand_uint32() >> (32 - k), rejection until r < n) over the public MT19937
// recurrence.  tests/test_golden_host.py holds it to random.shuffle itself (result AND state after).
#include <stdint.h>
#include "../../include/ader_hip.h"

namespace {
const int MT_N = 624, MT_M = 397;

inline void mt_twist(uint32_t* mt) {
    const uint32_t UP = 0x80000000u, LO = 0x7fffffffu, MAG = 0x9908b0dfu;
    int k = 0;
    for (; k < MT_N - MT_M; ++k) { const uint32_t y = (mt[k] & UP) | (mt[k + 1] & LO); mt[k] = mt[k + MT_M] ^ (y >> 1) ^ ((y & 1u) ? MAG : 0u); }
    for (; k < MT_N - 1; ++k) { const uint32_t y = (mt[k] & UP) | (mt[k + 1] & LO); mt[k] = mt[k + (MT_M - MT_N)] ^ (y >> 1) ^ ((y & 1u) ? MAG : 0u); }
    const uint32_t y = (mt[MT_N - 1] & UP) | (mt[0] & LO);
    mt[MT_N - 1] = mt[MT_M - 1] ^ (y >> 1) ^ ((y & 1u) ? MAG : 0u);
}
}  // namespace

extern "C" {

// random.shuffle(x) for an int64 array x[n] given the `random` module's state: mt_state[0..623] the Mersenne-Twister words,
// mt_state[624] the index (random.getstate()[1]); both are advanced in place exactly as CPython advances them.  n < 2^31.
int ader_host_shuffle(uint32_t* mt_state, int64_t* x, int64_t n) {
    if (!mt_state || (n > 0 && !x) || n < 0 || n >= (int64_t)1 << 31 || mt_state[MT_N] > (uint32_t)MT_N) return -2;
    int idx = (int)mt_state[MT_N];
    for (int64_t i = n - 1; i >= 1; --i) {
        const uint32_t m = (uint32_t)(i + 1);
        const int k = 32 - __builtin_clz(m);                    // (i + 1).bit_length()
        uint32_t r;
        do {
            if (idx >= MT_N) { mt_twist(mt_state); idx = 0; }
            uint32_t y = mt_state[idx++];
            y ^= y >> 11; y ^= (y << 7) & 0x9d2c5680u; y ^= (y << 15) & 0xefc60000u; y ^= y >> 18;
            r = y >> (32 - k);
        } while (r >= m);
        const int64_t t = x[i]; x[i] = x[r]; x[r] = t;
    }
    mt_state[MT_N] = (uint32_t)idx;
    return 0;
}

// Sampler rows of n sessions given as one flat item array (util.py:161-169, 226-227): row i = up to the last `maxlen` inputs of
// session i right-aligned in zeros, then its label (= last item); sessions shorter than 2 leave an all-zero row flagged invalid.
// flat: the sessions' items back to back, lens [n]; rows [n][maxlen + 1] must be ZERO on entry; valid [n] bytes.
int ader_host_pack_rows(const int32_t* flat, const int64_t* lens, int64_t n, int maxlen, int32_t* rows, unsigned char* valid) {
    if (n < 0 || maxlen < 1 || (n > 0 && (!flat || !lens || !rows || !valid))) return -2;
    const int64_t w = (int64_t)maxlen + 1;
    int64_t at = 0;
    for (int64_t i = 0; i < n; ++i) {
        const int64_t L = lens[i];
        if (L < 0) return -2;
        valid[i] = L > 1;
        if (L > 1) {
            const int64_t k = (L - 1 < maxlen) ? L - 1 : maxlen;
            const int32_t* src = flat + at + (L - 1 - k);
            int32_t* dst = rows + i * w + (maxlen - k);
            for (int64_t j = 0; j < k; ++j) dst[j] = src[j];
            rows[i * w + maxlen] = flat[at + L - 1];
        }
        at += L;
    }
    return 0;
}

// ... of sessions given as (start, length) pairs into a shared flat item array of flat_n items (ader_amd/data.py: PackedSessions -- a
// prefix of a session is the same start with a shorter length, a split is a gather of pairs; no item is copied before the rows are cut).
int ader_host_pack_rows_at(const int32_t* flat, int64_t flat_n, const int64_t* starts, const int64_t* lens, int64_t n, int maxlen,
                           int32_t* rows, unsigned char* valid) {
    if (n < 0 || maxlen < 1 || flat_n < 0 || (n > 0 && (!starts || !lens || !rows || !valid)) || (flat_n > 0 && !flat)) return -2;
    const int64_t w = (int64_t)maxlen + 1;
    for (int64_t i = 0; i < n; ++i) {
        const int64_t L = lens[i], at = starts[i];
        if (L < 0 || at < 0 || at + L > flat_n) return -2;
        valid[i] = L > 1;
        if (L > 1) {
            const int64_t k = (L - 1 < maxlen) ? L - 1 : maxlen;
            const int32_t* src = flat + at + (L - 1 - k);
            int32_t* dst = rows + i * w + (maxlen - k);
            for (int64_t j = 0; j < k; ++j) dst[j] = src[j];
            rows[i * w + maxlen] = flat[at + L - 1];
        }
    }
    return 0;
}

// ... of every session AND its prefixes down to length 2 (util.py:138-143: a session of length L yields itself, then s[:L-1], ...,
// s[:2]), in that order, straight from the flat item array -- the prefix lists themselves are not built (an evaluator needs only the
// rows).  rows [sum_i max(1, lens[i] - 1)][maxlen + 1] ZERO on entry; valid one byte per row.
int ader_host_prefix_rows(const int32_t* flat, const int64_t* lens, int64_t n, int maxlen, int32_t* rows, unsigned char* valid) {
    if (n < 0 || maxlen < 1 || (n > 0 && (!flat || !lens || !rows || !valid))) return -2;
    const int64_t w = (int64_t)maxlen + 1;
    int64_t at = 0, r = 0;
    for (int64_t i = 0; i < n; ++i) {
        const int64_t L = lens[i];
        if (L < 0) return -2;
        if (L < 2) { valid[r++] = 0; at += L; continue; }         // the session itself: an all-zero, invalid row
        for (int64_t p = L; p >= 2; --p, ++r) {                   // prefix lengths L, L - 1, ..., 2
            valid[r] = 1;
            const int64_t k = (p - 1 < maxlen) ? p - 1 : maxlen;
            const int32_t* src = flat + at + (p - 1 - k);
            int32_t* dst = rows + r * w + (maxlen - k);
            for (int64_t j = 0; j < k; ++j) dst[j] = src[j];
            rows[r * w + maxlen] = flat[at + p - 1];
        }
        at += L;
    }
    return 0;
}

}  // extern "C"

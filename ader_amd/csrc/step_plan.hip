// Native step driver: one C-ABI call enqueues a whole train step (include/ader_hip.h: ader_step_*).
//
// The reference issues ONE `sess.run(train_op)` per step (main.py:220-256): the graph executor, not Python, walks the ops.  Here a
// step is ~20 launcher calls on two HIP streams with event edges between them; driven one ctypes call at a time from Python the host
// needed 0.22-0.27 ms per step against 0.38 ms of GPU time at the shipped datasets' shapes.  A plan is the recorded launch sequence of
// one (shape, mode): per launch the launcher, its argument slots and the lane it goes to; per cross-lane dependency an event edge.
// Host-side descriptors (AderSeqFwd, AderDrop, pointer arrays ...) are COPIED into the plan at creation, so a plan owns everything
// it passes by host pointer.  Per step only three things change and are patched before the walk: the input pointers / scalars
// (AderStepPatch), the dropout keys (AderStepKey: key = f(seed, step, site), the host side of the counter spec in common.h) and the
// two stream handles.  Host code only: no kernel lives in this file.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include <type_traits>
#include <utility>
#include <vector>
#include "../../include/ader_hip.h"

namespace {

// ---------------------------------------------------------------- typed trampolines over the launchers' own prototypes
template <class T> inline T slot_as(uint64_t v) {
    if constexpr (std::is_pointer_v<T>) {
        return reinterpret_cast<T>(static_cast<uintptr_t>(v));
    } else if constexpr (std::is_same_v<T, float>) {
        uint32_t b = static_cast<uint32_t>(v);
        float f;
        memcpy(&f, &b, 4);
        return f;
    } else {
        return static_cast<T>(static_cast<int64_t>(v));     // int / unsigned / long / size_t: the slot holds the sign-extended value
    }
}
template <class... A, size_t... I> inline int call_slots(int (*fn)(A...), const uint64_t* s, std::index_sequence<I...>) {
    return fn(slot_as<A>(s[I])...);
}
template <class F, F fn> struct Tramp;
template <class... A, int (*fn)(A...)> struct Tramp<int (*)(A...), fn> {
    static constexpr int n_args = sizeof...(A);
    static int call(const uint64_t* s) { return call_slots(fn, s, std::index_sequence_for<A...>{}); }
};

struct FnEntry {
    const char* name;
    int (*call)(const uint64_t*);
    int n_args;
};
#define ADER_PLAN_FN(f) {#f, &Tramp<decltype(&f), &f>::call, Tramp<decltype(&f), &f>::n_args}
// every launcher a train step may contain (its last argument is the stream)
const FnEntry kFns[] = {
    ADER_PLAN_FN(ader_build_rowinfo),      ADER_PLAN_FN(ader_seq_fwd),             ADER_PLAN_FN(ader_seqp_fwd),
    ADER_PLAN_FN(ader_seq_pack_plan),      ADER_PLAN_FN(ader_sparse_lists_meta),   ADER_PLAN_FN(ader_sparse_lists),
    ADER_PLAN_FN(ader_lx3_fwd_img_lnf),    ADER_PLAN_FN(ader_lx3_fwd_kd_lnf),      ADER_PLAN_FN(ader_lx3_fwd_img),
    ADER_PLAN_FN(ader_lx3_fwd),            ADER_PLAN_FN(ader_lx3_fwd_kd),          ADER_PLAN_FN(ader_seq_bwd_ffn),
    ADER_PLAN_FN(ader_seq_bwd_qkv),        ADER_PLAN_FN(ader_seqp_bwd_ffn),        ADER_PLAN_FN(ader_seqp_bwd_qkv),
    ADER_PLAN_FN(ader_attn_last_bwd),      ADER_PLAN_FN(ader_attn_x3_bwd),         ADER_PLAN_FN(ader_attn_bwd),
    ADER_PLAN_FN(ader_attnp_bwd),          ADER_PLAN_FN(ader_attnp_last_bwd),      ADER_PLAN_FN(ader_x3_rep_image),
    ADER_PLAN_FN(ader_tab_update_x3),      ADER_PLAN_FN(ader_tab_update_x3_kd),    ADER_PLAN_FN(ader_lbf_sum),
    ADER_PLAN_FN(ader_reduce_slabs),       ADER_PLAN_FN(ader_reduce_slabs_batch),  ADER_PLAN_FN(ader_embed_bwd_rows),
    ADER_PLAN_FN(ader_pos_grad_packed),    ADER_PLAN_FN(ader_gemm_atb_x3_batch),   ADER_PLAN_FN(ader_gemm_atb_x3_batch_pk),
    ADER_PLAN_FN(ader_gemm_atb_x3),        ADER_PLAN_FN(ader_adam_step),           ADER_PLAN_FN(ader_wprep),
    ADER_PLAN_FN(ader_ln_bwd),             ADER_PLAN_FN(ader_fill),                ADER_PLAN_FN(ader_feed_step),
    ADER_PLAN_FN(ader_concat_i32),
};
constexpr int kNumFns = sizeof(kFns) / sizeof(kFns[0]);

inline uint32_t lowbias32_host(uint32_t x) {
    x ^= x >> 16; x *= 0x7FEB352DU; x ^= x >> 15; x *= 0x846CA68BU; x ^= x >> 16;
    return x;
}
// ader_amd/engine: dropout_key(seed, step, site) -- the key of a dropout site for one step
inline uint32_t dropout_key_host(uint32_t seed, uint32_t step, uint32_t site) {
    const uint32_t a = lowbias32_host(seed ^ 0x9E3779B9U);
    return lowbias32_host(a + step * 0x85EBCA6BU + site * 0xC2B2AE35U);
}

}  // namespace

struct AderStepPlan {
    std::vector<AderStepOp> ops;
    std::vector<std::vector<uint64_t>> blobs;      // 8-byte aligned copies of the host descriptors
    std::vector<size_t> blob_bytes;
    std::vector<AderStepPatch> patches;
    std::vector<AderStepKey> keys;
    std::vector<hipEvent_t> events;                // one per WAIT op, created at the first enqueue (a plan can be built without a device)
    int n_waits = 0;
    uint32_t seed = 0;
    int failed_op = -1;
    int device = 0;
};

extern "C" {

int ader_step_fn_index(const char* name) {
    for (int i = 0; i < kNumFns; ++i)
        if (strcmp(kFns[i].name, name) == 0) return i;
    return -1;
}

int ader_step_fn_args(int fn) { return (fn >= 0 && fn < kNumFns) ? kFns[fn].n_args : -1; }

int ader_step_plan_create(const AderStepOp* ops, int n_ops, const AderStepBlob* blobs, int n_blobs, const AderStepPatch* patches,
                          int n_patches, const AderStepKey* keys, int n_keys, unsigned seed, AderStepPlan** out) {
    if (!ops || !out || n_ops <= 0 || n_blobs < 0 || n_patches < 0 || n_keys < 0) return -2;
    AderStepPlan* p = new AderStepPlan();
    p->ops.assign(ops, ops + n_ops);
    p->seed = seed;
    (void)hipGetDevice(&p->device);
    int rc = 0;
    for (int i = 0; i < n_ops && !rc; ++i) {
        const AderStepOp& o = p->ops[i];
        if (o.kind == ADER_STEP_LAUNCH) {
            if (o.fn < 0 || o.fn >= kNumFns || o.n_args != kFns[o.fn].n_args || o.n_args > ADER_STEP_MAX_ARGS || (o.stream | 1) != 1) rc = -3;
        } else if (o.kind == ADER_STEP_WAIT) {
            if ((o.stream | 1) != 1 || (o.other | 1) != 1 || o.stream == o.other) rc = -3;
        } else {
            rc = -3;
        }
    }
    p->blobs.resize(n_blobs);
    for (int b = 0; b < n_blobs && !rc; ++b) {
        const AderStepBlob& s = blobs[b];
        if (s.op < 0 || s.op >= n_ops || p->ops[s.op].kind != ADER_STEP_LAUNCH || s.arg < 0 || s.arg >= p->ops[s.op].n_args - 1 || !s.src) { rc = -2; break; }
        p->blobs[b].assign((s.bytes + 7) / 8 + 1, 0);
        p->blob_bytes.push_back(s.bytes);
        memcpy(p->blobs[b].data(), s.src, s.bytes);
        p->ops[s.op].args[s.arg] = reinterpret_cast<uintptr_t>(p->blobs[b].data());
    }
    for (int i = 0; i < n_patches && !rc; ++i) {
        const AderStepPatch& q = patches[i];
        if (q.input < 0 || q.input >= ADER_STEP_MAX_INPUTS) rc = -2;
        else if (q.blob >= 0) { if (q.blob >= n_blobs || q.offset + 8 > p->blobs[q.blob].size() * 8) rc = -2; }
        else if (q.op < 0 || q.op >= n_ops || p->ops[q.op].kind != ADER_STEP_LAUNCH || q.arg < 0 || q.arg >= p->ops[q.op].n_args - 1) rc = -2;
    }
    for (int i = 0; i < n_keys && !rc; ++i)
        if (keys[i].blob < 0 || keys[i].blob >= n_blobs || keys[i].offset + 4 > p->blobs[keys[i].blob].size() * 8) rc = -2;
    if (!rc) {
        p->patches.assign(patches, patches + n_patches);
        p->keys.assign(keys, keys + n_keys);
        for (int i = 0; i < n_ops; ++i)
            if (p->ops[i].kind == ADER_STEP_WAIT) p->ops[i].fn = p->n_waits++;      // WAIT: index of its event (created at the first enqueue)
    }
    if (rc) {
        for (hipEvent_t ev : p->events) (void)hipEventDestroy(ev);
        delete p;
        return rc;
    }
    *out = p;
    return 0;
}

int ader_step_plan_destroy(AderStepPlan* p) {
    if (!p) return 0;
    for (hipEvent_t ev : p->events) (void)hipEventDestroy(ev);
    delete p;
    return 0;
}

static inline void apply_patches(AderStepPlan* p, const uint64_t* inputs, unsigned step) {
    for (const AderStepPatch& q : p->patches) {
        const uint64_t v = inputs[q.input] + (uint64_t)q.delta;
        if (q.blob >= 0) memcpy(reinterpret_cast<char*>(p->blobs[q.blob].data()) + q.offset, &v, 8);
        else p->ops[q.op].args[q.arg] = v;
    }
    for (const AderStepKey& k : p->keys) {
        const uint32_t key = dropout_key_host(p->seed, step, (uint32_t)k.site);
        memcpy(reinterpret_cast<char*>(p->blobs[k.blob].data()) + k.offset, &key, 4);
    }
}

int ader_step_enqueue(AderStepPlan* p, const uint64_t* inputs, int n_inputs, unsigned step, void* main_stream, void* side_stream) {
    if (!p || (n_inputs > 0 && !inputs) || n_inputs > ADER_STEP_MAX_INPUTS) return -2;
    for (const AderStepPatch& q : p->patches)
        if (q.input >= n_inputs) return -2;
    apply_patches(p, inputs, step);
    while ((int)p->events.size() < p->n_waits) {
        hipEvent_t ev;
        hipError_t e = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
        if (e != hipSuccess) return (int)e;
        p->events.push_back(ev);
    }
    void* lanes[2] = {main_stream, side_stream};
    const int n = (int)p->ops.size();
    for (int i = 0; i < n; ++i) {
        AderStepOp& o = p->ops[i];
        if (o.kind == ADER_STEP_LAUNCH) {
            o.args[o.n_args - 1] = reinterpret_cast<uintptr_t>(lanes[o.stream]);
            const int rc = kFns[o.fn].call(o.args);
            if (rc != 0) { p->failed_op = i; return rc; }
        } else {
            hipEvent_t ev = p->events[o.fn];
            hipError_t e = hipEventRecord(ev, (hipStream_t)lanes[o.other]);
            if (e == hipSuccess) e = hipStreamWaitEvent((hipStream_t)lanes[o.stream], ev, 0);
            if (e != hipSuccess) { p->failed_op = i; return (int)e; }
        }
    }
    return 0;
}

// the ops / descriptor bytes a step WOULD issue for these inputs, without issuing them (tests: a recorded Python-driven step is
// compared slot by slot with the plan's patched form).  ops_out [n_ops]; blob b's bytes are copied to blob_out[b] (NULL: skipped).
int ader_step_plan_peek(AderStepPlan* p, const uint64_t* inputs, int n_inputs, unsigned step, AderStepOp* ops_out, void* const* blob_out) {
    if (!p || !ops_out || n_inputs > ADER_STEP_MAX_INPUTS) return -2;
    for (const AderStepPatch& q : p->patches)
        if (q.input >= n_inputs) return -2;
    apply_patches(p, inputs, step);
    memcpy(ops_out, p->ops.data(), p->ops.size() * sizeof(AderStepOp));
    if (blob_out)
        for (size_t b = 0; b < p->blobs.size(); ++b)
            if (blob_out[b]) memcpy(blob_out[b], p->blobs[b].data(), p->blob_bytes[b]);
    return 0;
}

int ader_step_plan_failed_op(const AderStepPlan* p) { return p ? p->failed_op : -1; }

}  // extern "C"

// anakin_amd/csrc/api_capture.hip - op-list capture (saber_hip_capture_begin / _end): the C-ABI analogue of
// hipStreamBeginCapture one level up. A caller that owns an op loop of its own - the reference's Net<T,P,R>::prediction,
// framework/core/net/net.cpp:417-509, dispatching one Saber operator after the other through BaseFunc::operator() - runs that
// loop ONCE between begin and end; every capturable saber_hip_*_run call made by the calling thread is recorded (nothing is
// launched) and the result is an ordinary saber_hip_net: the list can then be optimised (saber_hip_net_optimize: fused eltwise
// epilogues, sibling pairs, conv + pooling, conv1x1 chains), autotuned, captured into a hipGraph and replayed with one call
// per forward pass instead of ~90 operator dispatches.
//
// Tensors are identified by the caller's device pointers. The reference's memory planner (MemoryScheduler) aliases edges
// whose lifetimes do not overlap onto a few buffers, so one pointer stands for MANY tensors over the course of a pass: the
// capture renames them the way SSA renames variables - a WRITE to a pointer starts a new tensor (a fresh slot of the net's
// arena), a READ refers to the newest tensor written there. A pointer that is read before anything wrote it is an input of
// the pass: it becomes a tensor bound to the caller's memory (saber_hip_net_bind_tensor). The caller binds the tensors it
// wants to see at its own addresses afterwards (the graph outputs: saber_hip_net_tensor_of_ptr + saber_hip_net_bind_tensor);
// every other edge lives in the arena, un-aliased, which is what lets the executor move writes across operator boundaries.
// In-place operators (the conv + sum post-op, RES_SUM_INPLACE) keep reading and writing ONE tensor.
// Whole tensors at buffer starts only: an access that overlaps a live tensor at a different base address fails the capture
// (SABER_HIP_UNIMPL from saber_hip_capture_end) and the caller keeps its own loop.
#include "api_internal.h"

#include <map>

namespace saber_api {

struct Capture {
    saber_hip_net* net = nullptr;
    struct Live { size_t bytes; int id; };
    std::map<uintptr_t, Live> live;      // base address -> newest tensor there
    bool failed = false;
    std::string why;
    int ops = 0;

    // Something the op list cannot express: the capture is marked failed (saber_hip_capture_end reports SABER_HIP_UNIMPL and
    // the reason) but the CALL returns success - the caller's loop is the reference's, whose SABER_CHECK around every
    // dispatch is fatal (framework/core/operator/operator.h), and a failed capture must leave that loop usable.
    int bad(const std::string& msg) {
        if (!failed) why = msg;
        failed = true;
        return SABER_HIP_OK;
    }
    // does [a, a + n) overlap a live tensor that starts somewhere else?
    bool foreign_overlap(uintptr_t a, size_t n, bool erase) {
        bool hit = false;
        auto it = live.lower_bound(a);
        if (it != live.begin()) {
            auto pv = std::prev(it);
            if (pv->first != a && pv->first + pv->second.bytes > a) {
                hit = true;
                if (erase) live.erase(pv);
            }
        }
        for (it = live.upper_bound(a); it != live.end() && it->first < a + n;) {
            hit = true;
            if (erase) it = live.erase(it);
            else ++it;
        }
        return hit;
    }
    int read(const void* p, size_t bytes) {
        if (!p) return -1;
        const uintptr_t a = (uintptr_t)p;
        auto it = live.find(a);
        if (it != live.end()) {
            if (bytes > it->second.bytes) {
                // an input of the pass may be seen first through a smaller view; a tensor written during the pass may not grow
                if (!net->tensor_ext[it->second.id]) return bad("a read is larger than the tensor written at that address"), -2;
                if (foreign_overlap(a, bytes, false)) return bad("overlapping tensors"), -2;
                it->second.bytes = bytes;
                net->tensor_bytes[it->second.id] = bytes;
            }
            return it->second.id;
        }
        if (foreign_overlap(a, bytes, false)) return bad("a read overlaps a tensor at another base address"), -2;
        const int id = saber_hip_net_add_tensor(net, bytes);      // read before written: an input of the pass
        net->tensor_ext[id] = const_cast<void*>(p);
        live[a] = {bytes, id};
        note(p, id);
        return id;
    }
    int write(void* p, size_t bytes) {
        if (!p) return -1;
        const uintptr_t a = (uintptr_t)p;
        foreign_overlap(a, bytes, true);       // whatever lived under the new tensor is gone (the planner reuses buffers)
        const int id = saber_hip_net_add_tensor(net, bytes);
        live[a] = {bytes, id};
        note(p, id);
        return id;
    }
    int readwrite(void* p, size_t bytes) {     // in place: the operator reads the tensor it then overwrites
        // A target nobody has written earlier in the pass becomes an INPUT bound to the caller's memory that every run of the list
        // accumulates into: legitimate for a caller who refills it per run (a single captured ConvEltwise), fatal for anything that runs
        // the list repeatedly on its own - saber_hip_net_autotune refuses such a net (saber_hip_net::inplace_external).
        // (round-5 advisor: a caller-owned input that an earlier op of the pass only READ is in `live` too, with tensor_ext set)
        if (p) {
            auto it = live.find((uintptr_t)p);
            if (it == live.end() || net->tensor_ext[it->second.id]) net->inplace_external = true;
        }
        return read(p, bytes);
    }
    void note(const void* p, int id) {
        for (auto& pr : net->captured_ptr)
            if (pr.first == p) { pr.second = id; return; }
        net->captured_ptr.push_back({p, id});
    }
};

thread_local Capture* g_capture = nullptr;

static size_t esz(int dt) { return dt == SABER_HIP_F32 || dt == SABER_HIP_S32 ? 4 : 1; }

int capture_unsupported(const char* what) { return g_capture->bad(std::string(what) + " cannot be captured"); }

int capture_conv(saber_hip_conv* op, const void* x, void* y, const void* res) {
    Capture& c = *g_capture;
    if (!op || !x || !y) return fail(SABER_HIP_INVALID_VALUE, "null argument");
    if (!op->weights_set) return fail(SABER_HIP_INVALID_VALUE, "set_weights not called");
    if (op->pair_k2 || op->gpool) return c.bad("pair / global-pooling convs are executor-level objects");
    const saber_hip_conv_desc& d = op->d;
    if (d.res_mode == SABER_HIP_RES_ELTWISE && !res) return fail(SABER_HIP_INVALID_VALUE, "residual tensor required");
    const size_t in_b = (size_t)d.n * d.h * d.w * d.c * esz(d.in_dtype);
    const int oh = (op->pool_fused || op->pool2) ? op->pool_oh : op->oh, ow = (op->pool_fused || op->pool2) ? op->pool_ow : op->ow;
    const size_t out_b = (size_t)d.n * oh * ow * d.k * esz(d.out_dtype);
    const int in = c.read(x, in_b);
    int r = -1;
    if (d.res_mode == SABER_HIP_RES_ELTWISE) {
        const size_t res_b = (d.res_stride > 1 ? (size_t)d.n * d.res_h * d.res_w * d.k : (size_t)d.n * op->oh * op->ow * d.k) *
                             esz(d.out_dtype);      // (the eltwise operands have the output's element size: s8 | s8 -> s8, f32 | f32 -> f32)
        r = c.read(res, res_b);
    }
    const int out = d.res_mode == SABER_HIP_RES_SUM_INPLACE ? c.readwrite(y, out_b) : c.write(y, out_b);
    if (c.failed || in < 0 || out < 0) return c.bad("bad tensor");
    const int idx = saber_hip_net_add_conv(c.net, op, in, out, r);
    if (idx < 0) return c.bad("saber_hip_net_add_conv failed");
    ++c.ops;
    return SABER_HIP_OK;
}

int capture_fc(saber_hip_fc* fc, const void* x, float* y, bool quantised_input) {
    Capture& c = *g_capture;
    if (!fc || !x || !y) return fail(SABER_HIP_INVALID_VALUE, "null argument");
    const size_t in_b = (size_t)fc->d.m * fc->d.k * (quantised_input ? 1 : esz(fc->d.in_dtype));
    const int in = c.read(x, in_b);
    const int out = c.write(y, (size_t)fc->d.m * fc->d.n * 4);
    if (c.failed || in < 0 || out < 0) return c.bad("bad tensor");
    const int idx = quantised_input ? saber_hip_net_add_fc_q(c.net, fc, in, out) : saber_hip_net_add_fc(c.net, fc, in, out);
    if (idx < 0) return c.bad("saber_hip_net_add_fc failed");
    ++c.ops;
    return SABER_HIP_OK;
}

int capture_stream_op(OpKind kind, const char* name, const int* p, int np, const float* f, int nf, size_t count, const void* in,
                      size_t in_bytes, const void* in2, size_t in2_bytes, void* out, size_t out_bytes, void* out2, size_t out2_bytes) {
    Capture& c = *g_capture;
    if (!in || !out) return fail(SABER_HIP_INVALID_VALUE, "null argument");
    NetOp o;
    o.kind = kind;
    o.name = name;
    for (int i = 0; i < np && i < 16; ++i) o.p[i] = p[i];
    for (int i = 0; i < nf && i < 6; ++i) o.f[i] = f[i];
    o.count = count;
    o.in = c.read(in, in_bytes);
    o.in2 = in2 ? c.read(in2, in2_bytes) : -1;
    // an elementwise operator may run in place (out == in): the new tensor then simply takes over the address
    o.out = c.write(out, out_bytes);
    o.out2 = out2 ? c.write(out2, out2_bytes) : -1;
    if (c.failed || o.in < 0 || o.out < 0 || o.in2 == -2) return c.bad("bad tensor");
    c.net->ops.push_back(std::move(o));
    ++c.ops;
    return SABER_HIP_OK;
}

}  // namespace saber_api

int saber_hip_capture_begin(void) {
    if (g_capture) return fail(SABER_HIP_INVALID_VALUE, "a capture is already open on this thread");
    Capture* c = new Capture();
    int rc = saber_hip_net_create(&c->net);
    if (rc) {
        delete c;
        return rc;
    }
    g_capture = c;
    return SABER_HIP_OK;
}

int saber_hip_capture_active(void) { return g_capture ? 1 : 0; }

int saber_hip_capture_end(saber_hip_net_t** out) {
    if (!g_capture) return fail(SABER_HIP_INVALID_VALUE, "no capture is open on this thread");
    Capture* c = g_capture;
    g_capture = nullptr;
    int rc = SABER_HIP_OK;
    if (c->failed) rc = fail(SABER_HIP_UNIMPL, "capture: " + c->why);
    else if (!out) rc = fail(SABER_HIP_INVALID_VALUE, "null argument");
    if (rc) {
        saber_hip_net_destroy(c->net);
        if (out) *out = nullptr;
    } else {
        *out = c->net;
    }
    delete c;
    return rc;
}

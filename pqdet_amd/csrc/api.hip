// Error plumbing and the batched op executor of libpqdet_hip.so.
#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include "common.h"

static thread_local char g_err[512] = "";
static thread_local int g_failed_op = -1;

void pq_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* pq_last_error(void) { return g_err; }
extern "C" int pq_last_failed_op(void) { return g_failed_op; }
extern "C" const char* pq_version(void) { return "pqdet_hip 0.1 (gfx950)"; }

// ---- launch log (tests): the set of (kernel expression | enclosing function) strings launched since it was switched on
#include <mutex>
#include <set>
#include <string>
bool pq_launch_log_on = false;
static std::mutex g_log_mu;
static std::set<std::string> g_log;
void pq_note_launch(const char* kernel, const char* where) {
    std::lock_guard<std::mutex> lk(g_log_mu);
    g_log.insert(std::string(kernel) + " | " + where);
}
extern "C" void pq_launch_log(int32_t enable) {
    std::lock_guard<std::mutex> lk(g_log_mu);
    g_log.clear();
    pq_launch_log_on = enable != 0;
}
extern "C" int64_t pq_launch_log_dump(char* buf, int64_t cap) {
    std::lock_guard<std::mutex> lk(g_log_mu);
    std::string all;
    for (const auto& e : g_log) { all += e; all += '\n'; }
    if (buf && cap > 0) {
        const size_t n = all.size() < (size_t)cap - 1 ? all.size() : (size_t)cap - 1;
        memcpy(buf, all.data(), n);
        buf[n] = 0;
    }
    return (int64_t)all.size() + 1;
}

static thread_local hipEvent_t g_stop_event = nullptr;
hipEvent_t pq_take_stop_event() {
    hipEvent_t e = g_stop_event;
    g_stop_event = nullptr;
    return e;
}

static int run_one(const pq_op& op, void* stream) {
    switch (op.code) {
        case PQ_OP_CONV1X1_FWD: return pq_conv1x1_fwd(&op.u.c1f, stream);
        case PQ_OP_CONV1X1_DGRAD: return pq_conv1x1_dgrad(&op.u.c1d, stream);
        case PQ_OP_CONV1X1_WGRAD: return pq_conv1x1_wgrad(&op.u.c1w, stream);
        case PQ_OP_DW_FWD: return pq_dw3x3_fwd(&op.u.dwf, stream);
        case PQ_OP_DW_DGRAD: return pq_dw3x3_dgrad(&op.u.dwd, stream);
        case PQ_OP_DW_WGRAD: return pq_dw3x3_wgrad(&op.u.dww, stream);
        case PQ_OP_STEM_FWD: return pq_stem3x3_fwd(&op.u.stf, stream);
        case PQ_OP_STEM_WGRAD: return pq_stem3x3_wgrad(&op.u.stw, stream);
        case PQ_OP_BN_FWD_FIN: return pq_bn_fwd_finalize(&op.u.bnf, stream);
        case PQ_OP_BN_BWD_FIN: return pq_bn_bwd_finalize(&op.u.bnb, stream);
        case PQ_OP_COPY_FWD: return pq_copy_fwd(&op.u.cp, stream);
        case PQ_OP_ROUTE_BWD: return pq_route_bwd(&op.u.rb, stream);
        case PQ_OP_DECODE: return pq_decode(&op.u.dec, stream);
        case PQ_OP_LOSS_FWD: return pq_yolo_loss_fwd(&op.u.loss, stream);
        case PQ_OP_LOSS_BWD: return pq_yolo_loss_bwd(&op.u.loss, stream);
        case PQ_OP_ADAM: return pq_adam_step(&op.u.adam, stream);
        case PQ_OP_FILL: return pq_fill(op.u.fill.dst, op.u.fill.value, op.u.fill.n, stream);
        case PQ_OP_DW_BWD: return pq_dw3x3_bwd(&op.u.dwb, stream);
        case PQ_OP_GCONV_FWD: return pq_gconv3x3_fwd(&op.u.gcf, stream);
        case PQ_OP_GCONV_DGRAD: return pq_gconv3x3_dgrad(&op.u.gcd, stream);
        case PQ_OP_GCONV_WGRAD: return pq_gconv3x3_wgrad(&op.u.gcw, stream);
        case PQ_OP_TRANSPOSE: return pq_transpose_batch(&op.u.tr, stream);
        case PQ_OP_QAT_WPREP: return pq_qat_weight_prep(&op.u.qw, stream);
        case PQ_OP_QAT_WGRAD: return pq_qat_weight_grad(&op.u.qw, stream);
        case PQ_OP_QACT_FWD: return pq_qact_fwd(&op.u.qa, stream);
        case PQ_OP_QACT_BWD: return pq_qact_bwd(&op.u.qb, stream);
        case PQ_OP_POOL_REDUCE: return pq_pool_reduce(&op.u.pr, stream);
        case PQ_OP_CONV1X1_BWD: return pq_conv1x1_bwd(&op.u.c1b, stream);
        case PQ_OP_BN_EVAL: return pq_bn_eval_batch(&op.u.bne, stream);
        case PQ_OP_LAYER_FWD: return pq_layer_fwd(&op.u.lay, stream);
        case PQ_OP_XDW_FWD: return pq_xdw_fwd(&op.u.xdf, stream);
        case PQ_OP_XSTAT_FWD: return pq_xstat_fwd(&op.u.xsf, stream);
        case PQ_OP_XDW_BWD: return pq_xdw_bwd(&op.u.xdb, stream);
        case PQ_OP_REDUCE_BATCH: return pq_reduce_batch(&op.u.red, stream);
        default:
            pq_set_error("pq_run_ops: unknown op code %d", op.code);
            return PQ_EINVAL;
    }
}

extern "C" int pq_host_device_pointer(void* host_ptr, void** device_ptr) {
    if (!host_ptr || !device_ptr) { pq_set_error("pq_host_device_pointer: null pointer"); return PQ_EINVAL; }
    void* d = nullptr;
    if (hipHostGetDevicePointer(&d, host_ptr, 0) != hipSuccess || !d) {
        (void)hipGetLastError();
        pq_set_error("pq_host_device_pointer: not a mapped pinned allocation");
        return PQ_EUNSUPPORTED;
    }
    *device_ptr = d;
    return PQ_OK;
}

extern "C" int pq_run_ops(const pq_op* ops, int32_t n, void* stream) {
    g_failed_op = -1;
    if (!ops || n < 0) { pq_set_error("pq_run_ops: bad arguments"); return PQ_EINVAL; }
    for (int i = 0; i < n; ++i) {
        const int rc = run_one(ops[i], stream);
        if (rc != PQ_OK) { g_failed_op = i; return rc; }
    }
    return PQ_OK;
}

// Side-stream ops (PQ_OPF_SIDE: weight gradients, they only feed the optimiser) wait for the main stream through an event.
// On this runtime every hipEventRecord on the main queue costs the NEXT main-stream kernel 5-8 us of start latency (rocprofv3
// kernel trace of a train step: 74 such gaps = 0.5 ms on the critical chain), so side ops are released in groups: they are
// parked in issue order and one fork event covers `group` of them (PQ_SIDE_GROUP; default 1 = an event per op: measured, larger
// groups lose more through the later start of the weight gradients (cold caches, less overlap) than the events cost).  Parking is
// safe because nothing on the main stream overwrites what a parked op reads (gradient buffers are per tensor and never
// reused inside a step; activations are read-only in the backward), and every call flushes and joins before it returns.
// both streams live on one device: a device-scope release is all the fork / join events need (the default is a system-scope
// fence at every record: 11.77 -> 11.61 ms per train step without it); the fork event is BOUND to the dispatch of the op in front
// of it where that op supports it (PQ_LAUNCH_BINDABLE).
static unsigned event_flags() { return hipEventDisableTiming | hipEventDisableSystemFence; }
static bool bind_events() { return true; }
static int side_group() { return 1; }

static int run_ops2_impl(const pq_op* ops, int32_t n, void* stream, void* side_stream, bool final_join) {
    g_failed_op = -1;
    if (!ops || n < 0) { pq_set_error("pq_run_ops2: bad arguments"); return PQ_EINVAL; }
    if (!side_stream || side_stream == stream) return pq_run_ops(ops, n, stream);
    static thread_local hipEvent_t ev[64];
    static thread_local int nev = 0, next = 0;
    if (nev == 0) {
        for (int i = 0; i < 64; ++i)
            if (hipEventCreateWithFlags(&ev[i], event_flags()) != hipSuccess) { pq_set_error("pq_run_ops2: event create failed"); return PQ_ELAUNCH; }
        nev = 64;
    }
    hipStream_t main_s = (hipStream_t)stream, side_s = (hipStream_t)side_stream;
    const int group = side_group();
    int parked[64];
    int nparked = 0;
    bool used_side = false;
    hipEvent_t bound = nullptr;          // fork event already bound to the last main-stream kernel (no record needed)
    bool main_dirty = true;              // a main-stream op has been issued since the last fork (else the side stream is already ordered)
    hipEvent_t early_join = nullptr;     // join event already recorded behind the last side op (see flush)
    auto flush = [&]() -> int {          // one fork event, then every parked op in issue order
        if (nparked == 0) return PQ_OK;
        bool need_fork = false;          // PQ_OPF_NOFORK ops are ordered by the side stream itself
        for (int k = 0; k < nparked; ++k) need_fork |= !(ops[parked[k]].flags & PQ_OPF_NOFORK);
        if (main_dirty && need_fork) {
            hipEvent_t e = bound;
            if (!e) { e = ev[next]; next = (next + 1) & 63; }
            if ((!bound && hipEventRecord(e, main_s) != hipSuccess) || hipStreamWaitEvent(side_s, e, 0) != hipSuccess) {
                pq_set_error("pq_run_ops2: fork failed"); g_failed_op = parked[0]; return PQ_ELAUNCH;
            }
            main_dirty = false;
        }
        bound = nullptr;
        const int last = parked[nparked - 1];
        // the LAST side op of a joined call (the stem's weight gradient at the end of a backward): the main stream has nothing
        // left to do but wait for it, so the lazy ops that rode along (small reductions of main-stream products) run THERE,
        // beside it, instead of in front of it on the side stream (4 x 5 us ahead of the step's last kernel)
        const bool tail = final_join && nparked > 1 && last == n - 1 && !(ops[last].flags & PQ_OPF_LAZY);
        for (int k = 0; k < nparked; ++k) {
            const bool on_main = tail && (ops[parked[k]].flags & PQ_OPF_LAZY) && !(ops[parked[k]].flags & PQ_OPF_NOFORK);
            const int rc = run_one(ops[parked[k]], on_main ? stream : side_stream);
            if (rc != PQ_OK) { g_failed_op = parked[k]; return rc; }
        }
        nparked = 0;
        used_side = true;
        // if the next thing that concerns the side stream is a JOIN (or the end of the call), record the join event NOW, right
        // behind the last side op: by the time the main stream gets there it is usually signalled already and the wait costs
        // nothing (recorded at the join itself, the record -> signal -> wait round trip through both queues left the main queue
        // idle for ~45 us in front of the last head's loss finalize)
        early_join = nullptr;
        for (int j = last + 1; j <= n; ++j) {
            if (j == n) { if (final_join) early_join = ev[next]; break; }
            if (ops[j].flags & PQ_OPF_SIDE) break;
            if (ops[j].flags & PQ_OPF_JOIN) { early_join = ev[next]; break; }
        }
        if (early_join) {
            next = (next + 1) & 63;
            if (hipEventRecord(early_join, side_s) != hipSuccess) { pq_set_error("pq_run_ops2: join record failed"); return PQ_ELAUNCH; }
        }
        return PQ_OK;
    };
    const bool bind = bind_events();
    bool side_dirty = false;             // side-stream work issued since the last join
    auto join = [&]() -> int {
        { const int rc = flush(); if (rc != PQ_OK) return rc; }
        if (used_side && side_dirty) {
            hipEvent_t e = early_join;
            if (!e) {
                e = ev[next]; next = (next + 1) & 63;
                if (hipEventRecord(e, side_s) != hipSuccess) { pq_set_error("pq_run_ops2: join failed"); return PQ_ELAUNCH; }
            }
            if (hipStreamWaitEvent(main_s, e, 0) != hipSuccess) { pq_set_error("pq_run_ops2: join failed"); return PQ_ELAUNCH; }
            early_join = nullptr;
            side_dirty = false;
        }
        return PQ_OK;
    };
    // place-holder slots the plan left behind (a fill of zero elements on the main stream: e.g. a BatchNorm finalize whose work a
    // neighbouring kernel took over) are invisible: nothing is launched for them and the fork look-ahead reads through them
    auto is_nop = [&](int k) { return ops[k].code == PQ_OP_FILL && ops[k].flags == 0 && ops[k].u.fill.n == 0; };
    for (int i = 0; i < n; ++i) {
        if (is_nop(i)) continue;
        if (ops[i].flags & PQ_OPF_SIDE) {
            parked[nparked++] = i;
            side_dirty = true;
            const bool lazy = (ops[i].flags & PQ_OPF_LAZY) != 0 && nparked < 60;      // rides with the next side op's fork
            if (!lazy && nparked >= group) { const int rc = flush(); if (rc != PQ_OK) return rc; }
            continue;
        }
        if (ops[i].flags & PQ_OPF_JOIN) { const int rc = join(); if (rc != PQ_OK) return rc; }
        // the op after this one releases the parked group: offer the fork event to this op's kernel launch
        int nx = i + 1;
        while (nx < n && is_nop(nx)) ++nx;
        const bool forks_next = bind && nx < n && (ops[nx].flags & PQ_OPF_SIDE) && !(ops[nx].flags & (PQ_OPF_LAZY | PQ_OPF_NOFORK)) && nparked + 1 >= group;
        bound = nullptr;
        if (forks_next) g_stop_event = ev[next];
        const int rc = run_one(ops[i], stream);
        main_dirty = true;
        if (forks_next) {
            if (g_stop_event == nullptr) { bound = ev[next]; next = (next + 1) & 63; }      // taken: the kernel carries it
            g_stop_event = nullptr;
        }
        if (rc != PQ_OK) { g_failed_op = i; return rc; }
    }
    if (!final_join) return flush();         // the caller orders whatever consumes the side stream's results itself
    return join();
}
extern "C" int pq_run_ops2(const pq_op* ops, int32_t n, void* stream, void* side_stream) {
    return run_ops2_impl(ops, n, stream, side_stream, true);
}
extern "C" int pq_run_ops2_nojoin(const pq_op* ops, int32_t n, void* stream, void* side_stream) {
    return run_ops2_impl(ops, n, stream, side_stream, false);
}

extern "C" int pq_run_ops_timed(const pq_op* ops, int32_t n, void* stream, float* ms) {
    g_failed_op = -1;
    if (!ops || n < 0 || !ms) { pq_set_error("pq_run_ops_timed: bad arguments"); return PQ_EINVAL; }
    static thread_local hipEvent_t* ev = nullptr;
    static thread_local int nev = 0;
    if (nev < n + 1) {
        for (int i = 0; i < nev; ++i) (void)hipEventDestroy(ev[i]);
        delete[] ev;
        ev = new hipEvent_t[n + 1];
        nev = n + 1;
        for (int i = 0; i < nev; ++i)
            if (hipEventCreate(&ev[i]) != hipSuccess) { pq_set_error("pq_run_ops_timed: hipEventCreate failed"); return PQ_ELAUNCH; }
    }
    hipStream_t st = (hipStream_t)stream;
    (void)hipEventRecord(ev[0], st);
    for (int i = 0; i < n; ++i) {
        const int rc = run_one(ops[i], stream);
        if (rc != PQ_OK) { g_failed_op = i; return rc; }
        (void)hipEventRecord(ev[i + 1], st);
    }
    if (hipEventSynchronize(ev[n]) != hipSuccess) { pq_set_error("pq_run_ops_timed: sync failed"); return PQ_ELAUNCH; }
    for (int i = 0; i < n; ++i) (void)hipEventElapsedTime(&ms[i], ev[i], ev[i + 1]);
    return PQ_OK;
}

extern "C" int pq_run_ops2_timed(const pq_op* ops, int32_t n, void* stream, void* side_stream, float* ms) {
    g_failed_op = -1;
    if (!ops || n < 0 || !ms) { pq_set_error("pq_run_ops2_timed: bad arguments"); return PQ_EINVAL; }
    if (!side_stream || side_stream == stream) return pq_run_ops_timed(ops, n, stream, ms);
    static thread_local hipEvent_t* ev = nullptr;
    static thread_local int nev = 0;
    const int need = 2 * n + 2;
    if (nev < need) {
        for (int i = 0; i < nev; ++i) (void)hipEventDestroy(ev[i]);
        delete[] ev;
        ev = new hipEvent_t[need];
        nev = need;
        for (int i = 0; i < nev; ++i)
            if (hipEventCreate(&ev[i]) != hipSuccess) { pq_set_error("pq_run_ops2_timed: hipEventCreate failed"); return PQ_ELAUNCH; }
    }
    hipStream_t main_s = (hipStream_t)stream, side_s = (hipStream_t)side_stream;
    hipEvent_t fork = ev[2 * n], join = ev[2 * n + 1];
    bool used_side = false, main_dirty = true, side_dirty = false;
    for (int i = 0; i < n; ++i) {
        const bool side = (ops[i].flags & PQ_OPF_SIDE) != 0;
        if (!side && (ops[i].flags & PQ_OPF_JOIN) && side_dirty) {
            if (hipEventRecord(join, side_s) != hipSuccess || hipStreamWaitEvent(main_s, join, 0) != hipSuccess) {
                pq_set_error("pq_run_ops2_timed: join failed"); g_failed_op = i; return PQ_ELAUNCH;
            }
            side_dirty = false;
        }
        if (side) {
            side_dirty = true;
            if (main_dirty && !(ops[i].flags & PQ_OPF_NOFORK)) {
                // one reusable fork event is enough here: hipStreamWaitEvent captures the record that precedes it
                if (hipEventRecord(fork, main_s) != hipSuccess || hipStreamWaitEvent(side_s, fork, 0) != hipSuccess) {
                    pq_set_error("pq_run_ops2_timed: fork failed"); g_failed_op = i; return PQ_ELAUNCH;
                }
                main_dirty = false;
            }
            used_side = true;
        } else {
            main_dirty = true;
        }
        hipStream_t st = side ? side_s : main_s;
        (void)hipEventRecord(ev[2 * i], st);
        const int rc = run_one(ops[i], st);
        if (rc != PQ_OK) { g_failed_op = i; return rc; }
        (void)hipEventRecord(ev[2 * i + 1], st);
    }
    if (used_side && side_dirty) {
        if (hipEventRecord(join, side_s) != hipSuccess || hipStreamWaitEvent(main_s, join, 0) != hipSuccess) {
            pq_set_error("pq_run_ops2_timed: join failed"); return PQ_ELAUNCH;
        }
    }
    if (hipStreamSynchronize(main_s) != hipSuccess) { pq_set_error("pq_run_ops2_timed: sync failed"); return PQ_ELAUNCH; }
    for (int i = 0; i < n; ++i) (void)hipEventElapsedTime(&ms[i], ev[2 * i], ev[2 * i + 1]);
    return PQ_OK;
}

// ---- BatchNorm statistics hand-over (pq_bn_stat)
int conv1x1_fwd_stat_ok(const pq_conv1x1_fwd_args* a, int role);     // gemm.hip
int dw_fwd_stat_ok(const pq_dw3x3_fwd_args* a, int role);           // dwtile.hip
int gconv_fwd_stat_ok(const pq_gconv3x3_fwd_args* a, int role);     // gconv.hip
extern "C" int32_t pq_bn_stat_ok(int32_t code, const void* args, int32_t role) {
    if (!args || (role != 0 && role != 1)) return 0;
    switch (code) {
        case PQ_OP_CONV1X1_FWD: return conv1x1_fwd_stat_ok((const pq_conv1x1_fwd_args*)args, role);
        case PQ_OP_DW_FWD: return dw_fwd_stat_ok((const pq_dw3x3_fwd_args*)args, role);
        case PQ_OP_GCONV_FWD: return gconv_fwd_stat_ok((const pq_gconv3x3_fwd_args*)args, role);
        case PQ_OP_COPY_FWD: {           // consumer only: shortcut / route / upsample read their operands through src_coef4
            const pq_copy_args* c = (const pq_copy_args*)args;
            return role == 1 && !c->a.qstate && !c->b.qstate && (c->C & 3) == 0;
        }
        case PQ_OP_XSTAT_FWD: {          // consumer only: the moments pass is the FIRST reader of a fused block's input
            const pq_xstat_fwd_args* c = (const pq_xstat_fwd_args*)args;
            return role == 1 && !c->in.qstate && (c->I == 16 || c->I == 24);
        }
        default: return 0;
    }
}
// ---- the same for the backward sums (pq_bn_bstat): role 0 = the call's sink can add into accumulators (pq_gsink.stat_acc),
// role 1 = the call derives its gradient operand's coefficients from them (its dy_stat)
int conv1x1_dgrad_bstat_ok(const pq_conv1x1_dgrad_args* a, int role);       // gemm.hip
int conv1x1_bwd_bstat_ok(const pq_conv1x1_bwd_args* a);           // pwbwd.hip
int dw_bwd_bstat_ok(const pq_dw3x3_bwd_args* a);                  // dwwalk.hip
static bool sink_has_stats(const pq_gsink& s) { return s.g && s.finalize && s.mean && s.partial && !s.qstate; }
extern "C" int32_t pq_bn_bstat_ok(int32_t code, const void* args, int32_t role) {
    if (!args || (role != 0 && role != 1)) return 0;
    switch (code) {
        case PQ_OP_CONV1X1_DGRAD: return conv1x1_dgrad_bstat_ok((const pq_conv1x1_dgrad_args*)args, role);
        case PQ_OP_CONV1X1_WGRAD: {       // consumer only: every kernel of the entry derives per lane (bias gradients, strided and QAT included)
            const pq_conv1x1_wgrad_args* a = (const pq_conv1x1_wgrad_args*)args;
            return role == 1 && a->dy.y != nullptr && a->dy.mu != nullptr;
        }
        case PQ_OP_CONV1X1_BWD: {
            const pq_conv1x1_bwd_args* a = (const pq_conv1x1_bwd_args*)args;
            return role == 0 ? conv1x1_bwd_bstat_ok(a) : (a->dy.y != nullptr && a->dy.mu != nullptr);
        }
        case PQ_OP_DW_BWD: {
            const pq_dw3x3_bwd_args* a = (const pq_dw3x3_bwd_args*)args;
            return role == 0 ? (a->part == 5 && sink_has_stats(a->dx)) : dw_bwd_bstat_ok(a);
        }
        case PQ_OP_ROUTE_BWD: {           // producer only; either sink (the caller sets stat_acc on the one(s) it means)
            const pq_route_bwd_args* a = (const pq_route_bwd_args*)args;
            return role == 0 && (sink_has_stats(a->dx) || sink_has_stats(a->dx2));
        }
        default: return 0;
    }
}
extern "C" int64_t pq_bn_stat_bytes(int32_t ldp, int32_t slots) { return (int64_t)(slots > 0 ? slots : 1) * 2 * ldp * 2 * 8; }
// replicas by the number of row blocks that add to one channel: ~64 same-address atomics in flight cost nothing, 512 cost 12 us
// (scripts/probes/bnstat_probe.hip); never more than 4, because EVERY workgroup of the consumer reads slots x 2 x 16 bytes per
// channel (16 replicas made the consumers' prologues cost more than the finalize launch they replace)
extern "C" int32_t pq_bn_stat_slots(int32_t nblk) { return nblk <= 96 ? 1 : 4; }

extern "C" int32_t pq_abi_sizeof(int32_t code) {
    switch (code) {
        case 0: return (int32_t)sizeof(pq_op);
        case 107: return (int32_t)sizeof(pq_bn_stat);
        case 108: return (int32_t)sizeof(pq_src);
        case 109: return (int32_t)sizeof(pq_bn_bstat);
        case PQ_OP_CONV1X1_FWD: return (int32_t)sizeof(pq_conv1x1_fwd_args);
        case PQ_OP_CONV1X1_DGRAD: return (int32_t)sizeof(pq_conv1x1_dgrad_args);
        case PQ_OP_CONV1X1_WGRAD: return (int32_t)sizeof(pq_conv1x1_wgrad_args);
        case PQ_OP_DW_FWD: return (int32_t)sizeof(pq_dw3x3_fwd_args);
        case PQ_OP_DW_DGRAD: return (int32_t)sizeof(pq_dw3x3_dgrad_args);
        case PQ_OP_DW_WGRAD: return (int32_t)sizeof(pq_dw3x3_wgrad_args);
        case PQ_OP_STEM_FWD: return (int32_t)sizeof(pq_stem3x3_fwd_args);
        case PQ_OP_STEM_WGRAD: return (int32_t)sizeof(pq_stem3x3_wgrad_args);
        case PQ_OP_BN_FWD_FIN: return (int32_t)sizeof(pq_bn_fwd_finalize_args);
        case PQ_OP_BN_BWD_FIN: return (int32_t)sizeof(pq_bn_bwd_finalize_args);
        case PQ_OP_COPY_FWD: return (int32_t)sizeof(pq_copy_args);
        case PQ_OP_ROUTE_BWD: return (int32_t)sizeof(pq_route_bwd_args);
        case PQ_OP_DECODE: return (int32_t)sizeof(pq_decode_args);
        case PQ_OP_LOSS_FWD: case PQ_OP_LOSS_BWD: return (int32_t)sizeof(pq_yolo_loss_args);
        case PQ_OP_ADAM: return (int32_t)sizeof(pq_adam_args);
        case PQ_OP_FILL: return (int32_t)sizeof(pq_fill_args);
        case PQ_OP_DW_BWD: return (int32_t)sizeof(pq_dw3x3_bwd_args);
        case PQ_OP_GCONV_FWD: return (int32_t)sizeof(pq_gconv3x3_fwd_args);
        case PQ_OP_GCONV_DGRAD: return (int32_t)sizeof(pq_gconv3x3_dgrad_args);
        case PQ_OP_GCONV_WGRAD: return (int32_t)sizeof(pq_gconv3x3_wgrad_args);
        case PQ_OP_TRANSPOSE: return (int32_t)sizeof(pq_transpose_args);
        case PQ_OP_QAT_WPREP: case PQ_OP_QAT_WGRAD: return (int32_t)sizeof(pq_qat_weight_args);
        case PQ_OP_QACT_FWD: return (int32_t)sizeof(pq_qact_args);
        case PQ_OP_QACT_BWD: return (int32_t)sizeof(pq_qact_bwd_args);
        case PQ_OP_POOL_REDUCE: return (int32_t)sizeof(pq_pool_reduce_args);
        case PQ_OP_XDW_FWD: return (int32_t)sizeof(pq_xdw_fwd_args);
        case PQ_OP_XSTAT_FWD: return (int32_t)sizeof(pq_xstat_fwd_args);
        case PQ_OP_XDW_BWD: return (int32_t)sizeof(pq_xdw_bwd_args);
        case PQ_OP_REDUCE_BATCH: return (int32_t)sizeof(pq_reduce_batch_args);
        case 110: return (int32_t)sizeof(pq_reduce_item);
        case PQ_OP_CONV1X1_BWD: return (int32_t)sizeof(pq_conv1x1_bwd_args);
        case PQ_OP_BN_EVAL: return (int32_t)sizeof(pq_bn_eval_args);
        case PQ_OP_LAYER_FWD: return (int32_t)sizeof(pq_layer_fwd_args);
        case 106: return (int32_t)sizeof(pq_bn_eval_item);
        case 105: return (int32_t)sizeof(pq_qat_weight_item);
        case 103: return (int32_t)sizeof(pq_transpose_item);
        case 104: return (int32_t)sizeof(pq_create_label_args);
        case 100: return (int32_t)sizeof(pq_recover_args);
        case 101: return (int32_t)sizeof(pq_nms_args);
        case 102: return (int32_t)sizeof(pq_decode_bwd_args);
        default: return -1;
    }
}

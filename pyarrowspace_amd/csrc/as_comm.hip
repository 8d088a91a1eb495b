// The two per-query exchange steps of a row-sharded search, issued by the library itself: RCCL all-gathers of the
// k-NN records and of the hit records on the query's own stream, and the whole staged search -- scan, exchange,
// lambda_q, scorer, exchange, merge, the escalation to the exact paths -- behind ONE host call
// (as_query_search_staged).  The host language above (pyarrowspace_amd/dist.py) used to issue the two collectives
// through torch.distributed: ~100 us of host time per query -- more than an 8-GPU scan (58 us at
// 1M x 768 / 8).
//
// RCCL is not linked: librccl is taken from the process (the copy torch.distributed has loaded, when it has) or from
// /opt/rocm/lib, by dlopen -- the library keeps working, single-GPU, on a box without it.
// Reference path this serves: PyArrowSpace::search, /root/reference/src/lib.rs:132-174 (one query per call).
#include <dlfcn.h>
#include <chrono>
#include <unistd.h>

#include <string>
#include <vector>

#include "as_query.hpp"

namespace {

typedef int nccl_result;
typedef struct ncclComm* nccl_comm;
struct nccl_unique_id {
    char internal[128];
};
constexpr int NCCL_CHAR = 0;   // ncclInt8 / ncclChar

struct Rccl {
    void* handle = nullptr;
    nccl_result (*get_unique_id)(nccl_unique_id*) = nullptr;
    nccl_result (*comm_init_rank)(nccl_comm*, int, nccl_unique_id, int) = nullptr;
    nccl_result (*all_gather)(const void*, void*, size_t, int, nccl_comm, hipStream_t) = nullptr;
    nccl_result (*comm_destroy)(nccl_comm) = nullptr;
    const char* (*error_string)(nccl_result) = nullptr;
};

Rccl* rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        const char* names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
        for (const char* nm : names)   // the copy already in the process first (torch's): two RCCLs in one process is one too many
            if ((r.handle = dlopen(nm, RTLD_NOW | RTLD_NOLOAD))) break;
        if (!r.handle)
            for (const char* nm : names)
                if ((r.handle = dlopen(nm, RTLD_NOW | RTLD_GLOBAL))) break;
        if (!r.handle) return;
        r.get_unique_id = (decltype(r.get_unique_id))dlsym(r.handle, "ncclGetUniqueId");
        r.comm_init_rank = (decltype(r.comm_init_rank))dlsym(r.handle, "ncclCommInitRank");
        r.all_gather = (decltype(r.all_gather))dlsym(r.handle, "ncclAllGather");
        r.comm_destroy = (decltype(r.comm_destroy))dlsym(r.handle, "ncclCommDestroy");
        r.error_string = (decltype(r.error_string))dlsym(r.handle, "ncclGetErrorString");
        if (!r.get_unique_id || !r.comm_init_rank || !r.all_gather || !r.comm_destroy) r.handle = nullptr;
    });
    return r.handle ? &r : nullptr;
}

as_status nccl_check(nccl_result res, const char* what) {
    if (res == 0) return AS_OK;
    Rccl* r = rccl();
    as::set_err("%s: RCCL error %d (%s)", what, res, r && r->error_string ? r->error_string(res) : "?");
    return AS_EHIP;
}

}  // namespace

struct as_comm {
    nccl_comm comm = nullptr;
    int rank = 0, world = 1, device = 0;
};

using namespace as;

extern "C" {

int32_t as_comm_available(void) { return rccl() ? 1 : 0; }

as_status as_comm_unique_id(void* out_128_bytes) {
    Rccl* r = rccl();
    if (!r || !out_128_bytes) {
        set_err("as_comm_unique_id: %s", r ? "null argument" : "librccl not found");
        return r ? AS_EINVAL : AS_EUNSUPPORTED;
    }
    nccl_unique_id id;
    AS_TRY(nccl_check(r->get_unique_id(&id), "ncclGetUniqueId"));
    memcpy(out_128_bytes, &id, sizeof(id));
    return AS_OK;
}

as_status as_comm_create(const void* id_128_bytes, int32_t rank, int32_t world, int32_t device, as_comm** out) {
    Rccl* r = rccl();
    if (!r || !id_128_bytes || !out || rank < 0 || rank >= world) {
        set_err("as_comm_create: %s", r ? "bad argument" : "librccl not found");
        return r ? AS_EINVAL : AS_EUNSUPPORTED;
    }
    AS_HIP(hipSetDevice(device));
    nccl_unique_id id;
    memcpy(&id, id_128_bytes, sizeof(id));
    as_comm* c = new as_comm();
    c->rank = rank;
    c->world = world;
    c->device = device;
    // (RCCL greets its first communicator with a version banner on stdout: a host that prints its results there -- bench.py's one
    // JSON line -- must not find it in between: the banner goes to stderr)
    fflush(stdout);
    const int saved = dup(1);
    if (saved >= 0) dup2(2, 1);
    const as_status s = nccl_check(r->comm_init_rank(&c->comm, world, id, rank), "ncclCommInitRank");   // collective: every rank calls it
    if (saved >= 0) {
        fflush(stdout);
        dup2(saved, 1);
        close(saved);
    }
    if (s != AS_OK) {
        delete c;
        return s;
    }
    *out = c;
    return AS_OK;
}

void as_comm_free(as_comm* c) {
    if (!c) return;
    Rccl* r = rccl();
    if (r && c->comm) r->comm_destroy(c->comm);
    delete c;
}

// The query's records are exchanged by the library from now on: gather buffers for `world` ranks' records.
as_status as_query_set_comm(as_query* q, as_comm* c) {
    if (!q || !c) {
        set_err("as_query_set_comm: null argument");
        return AS_EINVAL;
    }
    if (q->cap != 1 || !q->own_records) {
        set_err("as_query_set_comm: a single-query workspace with its own record buffers is required");
        return AS_EINVAL;
    }
    if ((int64_t)c->world * q->k > REC_CAP || (int64_t)c->world * (q->topk + 1) > HIT_CAP) {
        set_err("as_query_set_comm: %d ranks x (k = %lld, topk = %lld) exceed the merge capacities (%d, %d)", c->world, (long long)q->k,
                (long long)q->topk, REC_CAP, HIT_CAP);
        return AS_EUNSUPPORTED;
    }
    AS_HIP(hipSetDevice(q->sp->device));
    if (q->knn_all) hipFree(q->knn_all);
    if (q->hits_all) hipFree(q->hits_all);
    q->knn_all = nullptr;
    q->hits_all = nullptr;
    AS_HIP(hipMalloc(&q->knn_all, sizeof(as_knn_rec) * std::max<int64_t>(q->k, 1) * c->world));
    AS_HIP(hipMalloc(&q->hits_all, sizeof(as_hit_rec) * (q->topk + 1) * c->world));
    // the one-exchange pass: this rank's block (k-NN records, header, finished scorer candidates) and the gathered blocks
    if (q->xsend) hipFree(q->xsend);
    if (q->xall) hipFree(q->xall);
    q->xsend = q->xall = nullptr;
    const int64_t xb = as_query_x1_bytes(q, c->world);
    AS_HIP(hipMalloc(&q->xsend, (size_t)xb));
    AS_HIP(hipMalloc(&q->xall, (size_t)xb * c->world));
    q->comm = c;
    return AS_OK;
}

// One row-sharded search, every rank with the same query: this rank scans its rows [row_begin, row_end) of its space,
// the k nearest-neighbour records and the hit records of all ranks are all-gathered (RCCL, on the query's stream, no
// host wait in between) and merged identically everywhere.  The escalation is pyarrowspace_amd.dist.next_mode's: every
// rank sees the same merged flags, so every rank takes the same step -- at most four passes.
as_status as_query_search_staged(as_query* q, const double* query_host, int64_t d, int64_t row_begin, int64_t row_end, double tau,
                                 int64_t* out_idx, double* out_score, int64_t* out_len, double* out_lambda_q) {
    Rccl* r = rccl();
    if (!q || !q->comm || !r || !query_host || !out_idx || !out_score || !out_len) {
        set_err("as_query_search_staged: null argument or no communicator (as_query_set_comm)");
        return AS_EINVAL;
    }
    as_comm* c = q->comm;
    const int64_t krec = std::max<int64_t>(q->k, 1), hrec = q->topk + 1;
    int mode = 0;
    as_status st = AS_OK;
    // scan-side scorer candidates (as on one GPU): a query whose candidates overflow on some rank is rerun
    // without them -- every rank reads the same merged flag -- and the following 63 queries do not try
    bool sc = !(q->sc_crowded > 0 && (q->sc_crowded++ & 63) != 0);
    bool x1_fine = false;   // the one-exchange pass has been tried with the coarse scan allowed and some rank's candidates did not fit
    as_query_set_coarse(q, 1);
    for (int pass = 0; pass < 10; ++pass) {
        as_query_set_exact(q, mode);
        if (sc && mode == 0 && q->xsend && as_query_x1_usable(q, tau)) {
            // ONE exchange: every rank's k-NN records and finished scorer candidates (exact cosine, lambda) in one all-gather;
            // lambda_q, the scores and the ranking redundantly on every rank (as_search.hip, staged_x1_kernel).  3 launches
            // behind the scan and one collective, against 5 and two.
            const int64_t xb = as_query_x1_bytes(q, c->world);
            static const bool timing = getenv("ARROWSPACE_DEBUG") != nullptr;   // host time of the pass's three parts, every 100 passes
            static double t_acc[3] = {0, 0, 0};
            static int t_n = 0;
            auto now = [] { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
            const double t0 = timing ? now() : 0.0;
            as_status local = as_query_x1_begin(q, query_host, d, row_begin, row_end, tau, q->xsend, c->world);
            std::string local_msg = local != AS_OK ? err_slot() : std::string();
            if (local != AS_OK) (void)hipMemsetAsync(q->xsend, 0xff, (size_t)xb, q->stream);   // a failed rank's block: no records, flag bits all set
            const double t1 = timing ? now() : 0.0;
            AS_TRY(nccl_check(r->all_gather(q->xsend, q->xall, (size_t)xb, NCCL_CHAR, c->comm, q->stream), "ncclAllGather (one-exchange block)"));
            const double t2 = timing ? now() : 0.0;
            st = as_query_x1_finish(q, q->xall, c->world, tau, out_idx, out_score, out_len, out_lambda_q);
            if (timing) {
                t_acc[0] += t1 - t0;
                t_acc[1] += t2 - t1;
                t_acc[2] += now() - t2;
                if (++t_n == 100) {
                    dbg("one-exchange pass, host us: scan + block kernels enqueued %.1f, all-gather enqueued %.1f, finish kernel + wait %.1f",
                        t_acc[0] / 100, t_acc[1] / 100, t_acc[2] / 100);
                    t_acc[0] = t_acc[1] = t_acc[2] = 0.0;
                    t_n = 0;
                }
            }
            if (local != AS_OK) {
                set_err("%s", local_msg.c_str());
                return local;
            }
            if (st != AS_OK && st != AS_EZEROLAMBDA) return st;
            if (q->hout->overflow & 8) {
                set_err("as_query_search_staged: another rank of the index failed in this pass");
                return AS_EHIP;
            }
            if ((q->hout->overflow & 4) && !x1_fine) {
                // (a coarse scan's wider windows, on whichever rank: once more on the two-digit image -- every rank reads the same flag)
                x1_fine = true;
                as_query_set_coarse(q, 0);
                continue;
            }
            q->sc_crowded = (q->hout->overflow & 4) ? 1 : 0;
            if ((q->hout->overflow & 4) && debug_enabled()) {
                int hd[4] = {0, 0, 0, 0};
                (void)hipMemcpy(hd, q->xall + sizeof(as_knn_rec) * std::max<int64_t>(q->k, 1), sizeof(hd), hipMemcpyDeviceToHost);
                dbg("one-exchange pass: redo (rank 0's block: %d candidates, flags %d, largest share of a candidate block %d, blocks with an overflowed report %d; merged overflow bits %d)",
                    hd[0], hd[1], hd[2], hd[3], q->hout->overflow);
            }
            if (q->hout->overflow & 4) {   // some rank's candidates did not fit (or it had none to offer): the two-exchange chain
                sc = false;
                continue;
            }
            int32_t ki = 0, si = 0;
            as_query_flags(q, &ki, &si);
            const int overflow = (ki & 2) ? 1 : 0;
            if (!(ki & 1) && !overflow) return st;
            mode = (ki & 1) ? (overflow ? 3 : 1) : 4;
            continue;
        }
        q->staged_tau = sc && mode == 0 ? tau : -1.0;
        // A failure of THIS rank between the collectives (an allocation of the fp64 escalation, a wait that times out) must
        // not leave the peers blocked in an all-gather it never enters: the pass goes on -- both collectives are issued --,
        // the trailing hit record carries an error flag every rank merges, and every rank returns after the pass.  (A failing
        // collective itself is the communicator's failure: nothing further can be exchanged over it.)
        as_status local = as_query_scan(q, query_host, d, row_begin, row_end);
        q->staged_tau = -1.0;
        std::string local_msg = local != AS_OK ? err_slot() : std::string();
        AS_TRY(nccl_check(r->all_gather(q->knn, q->knn_all, sizeof(as_knn_rec) * krec, NCCL_CHAR, c->comm, q->stream), "ncclAllGather (k-NN records)"));
        if (local == AS_OK && (local = as_query_lambda(q, q->knn_all, krec * c->world)) != AS_OK) local_msg = err_slot();
        if (local == AS_OK && (local = as_query_score(q, tau)) != AS_OK) local_msg = err_slot();
        if (local != AS_OK) {
            // no hits of this rank (whatever the buffer holds is not this query's), the flag in the trailing record
            std::vector<as_hit_rec> fail((size_t)hrec);
            for (auto& h : fail) {
                h.idx = -1;
                h.score = 0.0;
            }
            fail[(size_t)hrec - 1].idx = -2;
            fail[(size_t)hrec - 1].score = 32.0;
            (void)hipMemcpyAsync(q->hits, fail.data(), sizeof(as_hit_rec) * hrec, hipMemcpyHostToDevice, q->stream);
            (void)hipStreamSynchronize(q->stream);   // (the source is this frame's)
        }
        AS_TRY(nccl_check(r->all_gather(q->hits, q->hits_all, sizeof(as_hit_rec) * hrec, NCCL_CHAR, c->comm, q->stream), "ncclAllGather (hit records)"));
        st = as_query_finish(q, q->hits_all, hrec * c->world, out_idx, out_score, out_len, out_lambda_q);
        q->staged_sc = 0;
        if (local != AS_OK) {
            as_query_set_exact(q, 0);
            set_err("%s", local_msg.c_str());
            return local;
        }
        if (q->hout->overflow & 8) {
            as_query_set_exact(q, 0);
            set_err("as_query_search_staged: another rank of the index failed in this pass");
            return AS_EHIP;
        }
        if (st != AS_OK && st != AS_EZEROLAMBDA) return st;
        if (sc && mode == 0) {
            q->sc_crowded = (q->hout->overflow & 4) ? 1 : 0;
            if (q->hout->overflow & 4) {   // some rank's candidates did not fit: the same pass again, on the threshold chain
                sc = false;
                continue;
            }
        }
        int32_t ki = 0, si = 0;
        as_query_flags(q, &ki, &si);
        const bool inexact = (ki & 1) || (si & 1);
        const int overflow = ((ki & 2) ? 1 : 0) | ((si & 2) ? 2 : 0);   // bit0: k-NN buffer, bit1: scorer buffer
        int next = -1;
        if (inexact && !(mode & 1)) next = (overflow || (mode & 2)) ? 3 : 1;
        else if ((overflow & 1) && !(mode & 6)) next = mode | 4;
        else if (overflow && !(mode & 2)) next = (mode | 2) & ~4;
        if (next < 0) {
            as_query_set_exact(q, 0);
            return st;
        }
        mode = next;
    }
    set_err("sharded search did not settle on an exact answer");
    return AS_EHIP;
}

}  // extern "C"

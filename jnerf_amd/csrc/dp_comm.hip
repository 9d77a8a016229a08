// Ray-batch data parallelism inside the library (the reference has no collective call sites at all - python/jnerf/utils/general.py:39-40 is its only
// multi-process hook).  Host code only: RCCL is bound at run time (dlopen of librccl.so.1 - the copy PyTorch-ROCm already mapped when the host side is torch, the
// ROCm one otherwise), so libngp_hip.so has no link-time dependency on it and single-GPU users never load it.
//
// Exchange step of one iteration, N ranks (one process per GPU), all on the caller's stream, in order - every arrow is a true data dependency, so a second stream
// would buy nothing but event packets (csrc/train_step.hip has the optional overlapped variant for the one place where there is slack):
//   backward (hash scatter overwrites the table gradient)
//   -> ONE RCCL group: reduce-scatter of the table gradient (every rank ends up with the SUM of its 1/N shard), all-reduce of the ragged tail (< 8 N floats)
//      and of the 10240 MLP weight gradients
//   -> fused Adam+EMA sweep of the rank's shard only (+ tail + MLP pack, replicated): 1/N of the 28 B/parameter stream
//   -> ONE RCCL group: all-gather of the updated shard of whatever the kernels read (fp32 table | fp16 shadow).
// xGMI is a full mesh of point-to-point links: a reduce-scatter moves S/N bytes over each of the N-1 links of a GPU concurrently, so the bytes per LINK shrink
// with N (DESIGN.md §5 has the resulting step-time model for N = 1, 2, 4, 8).
#include "ngp_common.h"
#include <dlfcn.h>
#include <mutex>
#include <string.h>

namespace {
// the slice of rccl.h this file needs (ncclResult_t 0 = success; ncclDataType_t: 6 = half, 7 = float; ncclRedOp_t 0 = sum)
typedef struct { char internal[NGP_COMM_ID_BYTES]; } RcclId;
struct Rccl {
	void *handle = nullptr;
	int (*GetUniqueId)(RcclId *) = nullptr;
	int (*CommInitRank)(void **, int, RcclId, int) = nullptr;
	int (*CommDestroy)(void *) = nullptr;
	int (*CommAbort)(void *) = nullptr;
	int (*AllReduce)(const void *, void *, size_t, int, int, void *, hipStream_t) = nullptr;
	int (*ReduceScatter)(const void *, void *, size_t, int, int, void *, hipStream_t) = nullptr;
	int (*AllGather)(const void *, void *, size_t, int, void *, hipStream_t) = nullptr;
	int (*GroupStart)() = nullptr;
	int (*GroupEnd)() = nullptr;
	const char *(*GetErrorString)(int) = nullptr;
	bool ok = false;
	char why[256] = "";
};
Rccl g_rccl;
std::once_flag g_rccl_once;

void load_rccl() {
	const char *names[3] = {getenv("NGP_RCCL_PATH"), "librccl.so.1", "librccl.so"};
	for (int i = 0; i < 3 && !g_rccl.handle; ++i)
		if (names[i] && names[i][0]) g_rccl.handle = dlopen(names[i], RTLD_NOW | RTLD_GLOBAL);
	if (!g_rccl.handle) { snprintf(g_rccl.why, sizeof(g_rccl.why), "cannot load librccl.so.1 (%s); set NGP_RCCL_PATH", dlerror()); return; }
#define SYM(field, name) do { *(void **)(&g_rccl.field) = dlsym(g_rccl.handle, name); if (!g_rccl.field) { snprintf(g_rccl.why, sizeof(g_rccl.why), "librccl: symbol %s missing", name); return; } } while (0)
	SYM(GetUniqueId, "ncclGetUniqueId"); SYM(CommInitRank, "ncclCommInitRank"); SYM(CommDestroy, "ncclCommDestroy"); SYM(CommAbort, "ncclCommAbort"); SYM(AllReduce, "ncclAllReduce");
	SYM(ReduceScatter, "ncclReduceScatter"); SYM(AllGather, "ncclAllGather"); SYM(GroupStart, "ncclGroupStart"); SYM(GroupEnd, "ncclGroupEnd"); SYM(GetErrorString, "ncclGetErrorString");
#undef SYM
	g_rccl.ok = true;
}
bool rccl_ready() { std::call_once(g_rccl_once, load_rccl); if (!g_rccl.ok) ngp_set_error("RCCL: %s", g_rccl.why); return g_rccl.ok; }
int nccl_dtype(int dtype) { return dtype == NGP_F16 ? 6 : 7; }
}  // namespace

struct NgpComm { void *nccl; int rank, world; };

#define RCCL_CALL(call, what) do { int r_ = (call); if (r_ != 0) { ngp_set_error("%s: RCCL error %d (%s)", what, r_, g_rccl.GetErrorString ? g_rccl.GetErrorString(r_) : "?"); return 1000 + r_; } } while (0)
// Inside an open group an error must not return past ncclGroupEnd: the thread's group would stay open and the next RCCL call of this process - the library's or
// torch's - would be queued into it silently.  RcclGroup opens a group and closes it on every path out of the scope; GROUP_CALL records the first
// failure, stops issuing, and the function returns it after the group was closed.
namespace {
struct RcclGroup {
	int rc = 0; bool open = false;
	RcclGroup() { rc = g_rccl.GroupStart(); open = rc == 0; }
	int close() { if (open) { open = false; const int r = g_rccl.GroupEnd(); if (!rc) rc = r; } return rc; }
	~RcclGroup() { close(); }
};
int rccl_fail(int r, const char *what) { ngp_set_error("%s: RCCL error %d (%s)", what, r, g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "?"); return 1000 + r; }
}  // namespace
#define GROUP_CALL(grp, call, what) do { if (!(grp).rc) { const int r_ = (call); if (r_ != 0) { (grp).rc = r_; rccl_fail(r_, what); } } } while (0)
#define GROUP_FINISH(grp, what) do { const bool failed_inside_ = (grp).rc != 0; const int r_ = (grp).close(); if (r_ != 0) { if (!failed_inside_) rccl_fail(r_, what); return 1000 + r_; } } while (0)

NGP_API int ngp_comm_unique_id(void *id_out_host) {
	NGP_REQUIRE(id_out_host, NGP_E_ARG, "ngp_comm_unique_id: null output");
	if (!rccl_ready()) return NGP_E_ARG;
	RcclId id; memset(&id, 0, sizeof(id));
	RCCL_CALL(g_rccl.GetUniqueId(&id), "ncclGetUniqueId");
	memcpy(id_out_host, &id, sizeof(id));
	return 0;
}

NGP_API int ngp_comm_init(void **comm_out, int rank, int world, const void *id_host) {
	NGP_REQUIRE(comm_out && id_host, NGP_E_ARG, "ngp_comm_init: null pointer");
	NGP_REQUIRE(world >= 1 && rank >= 0 && rank < world, NGP_E_ARG, "ngp_comm_init: rank %d of world %d", rank, world);
	if (!rccl_ready()) return NGP_E_ARG;
	RcclId id; memcpy(&id, id_host, sizeof(id));
	void *nc = nullptr;
	RCCL_CALL(g_rccl.CommInitRank(&nc, world, id, rank), "ncclCommInitRank");            // binds to the calling thread's current HIP device
	NgpComm *c = new NgpComm{nc, rank, world};
	*comm_out = c;
	return 0;
}

NGP_API int ngp_comm_destroy(void *comm) {
	if (!comm) return 0;
	NgpComm *c = (NgpComm *)comm;
	int rc = 0;
	if (c->nccl && g_rccl.ok) rc = g_rccl.CommDestroy(c->nccl);
	delete c;
	if (rc) { ngp_set_error("ncclCommDestroy: RCCL error %d", rc); return 1000 + rc; }
	return 0;
}

// ncclCommAbort: frees the communicator AND terminates its kernels that are still in flight (a collective whose peers never arrived spins on the device for ever and
// would block every later device-wide synchronisation of the process - ADVICE r4).  For a communicator that failed its self-test; a healthy one is destroyed.
NGP_API int ngp_comm_abort(void *comm) {
	if (!comm) return 0;
	NgpComm *c = (NgpComm *)comm;
	int rc = 0;
	if (c->nccl && g_rccl.ok) rc = g_rccl.CommAbort(c->nccl);
	delete c;
	if (rc) { ngp_set_error("ncclCommAbort: RCCL error %d", rc); return 1000 + rc; }
	return 0;
}

NGP_API int ngp_comm_rank_world(void *comm, int *rank, int *world) {
	NGP_REQUIRE(comm, NGP_E_ARG, "ngp_comm_rank_world: null communicator");
	if (rank) *rank = ((NgpComm *)comm)->rank;
	if (world) *world = ((NgpComm *)comm)->world;
	return 0;
}

// ---------------------------------------------------------------------------------------------------------------- how the table is dealt to the ranks (pure host arithmetic)
NGP_API int ngp_dp_plan(const uint32_t *level_table_host, uint64_t n_params, int world, int rank, int n_buckets, NgpDpPlan *out) {
	NGP_REQUIRE(out && level_table_host, NGP_E_ARG, "ngp_dp_plan: null pointer");
	NGP_REQUIRE(world >= 1 && rank >= 0 && rank < world, NGP_E_ARG, "ngp_dp_plan: rank %d of world %d", rank, world);
	NGP_REQUIRE(n_buckets == 1 || n_buckets == 2, NGP_E_ARG, "ngp_dp_plan: %d buckets (1 or 2)", n_buckets);
	memset(out, 0, sizeof(*out));
	const uint64_t A = 8ull * (uint64_t)world;                  // shard boundaries are multiples of 8 elements: 16-byte vectors of fp16 (wire) and fp32 (sweep) alike
	const uint64_t main_end = n_params / A * A;
	uint64_t cut1 = 0;
	if (n_buckets == 2) {                                       // the bucket boundary is where the scatter's coarse (run-combined) levels end: their accumulate launch finishes first
		for (int l = 0; l < 16; ++l) if (level_table_host[4 * l + 2] > NGP_DP_COARSE_RES_MAX) { cut1 = 2ull * level_table_host[4 * l]; out->cut_level = l; break; }
		cut1 = cut1 / A * A;                                    // rounded DOWN: every element below it belongs to a finished level
		if (cut1 == 0 || cut1 >= main_end) cut1 = 0;
	}
	out->n_buckets = cut1 ? 2 : 1;
	if (!cut1) out->cut_level = 16;
	out->cut[0] = 0; out->cut[1] = cut1 ? cut1 : main_end; out->cut[2] = main_end;
	for (uint32_t b = 0; b < out->n_buckets; ++b) {
		const uint64_t cnt = out->cut[b + 1] - out->cut[b];
		out->shard_count[b] = cnt / (uint64_t)world;
		out->shard_begin[b] = out->cut[b] + (uint64_t)rank * out->shard_count[b];
	}
	out->tail_begin = main_end; out->tail_count = n_params - main_end;
	out->world = world; out->rank = rank;
	return 0;
}

// ---------------------------------------------------------------------------------------------------------------- collectives (asynchronous on `stream`)
// §8b's export: SUM all-reduce, in place, of a list of gradient buffers - one RCCL group = one launch.  What the module (autograd) path of a data-parallel run calls.
NGP_API int ngp_allreduce_grads(void *comm, void *stream, int n_bufs, void *const *bufs_host, const uint64_t *counts_host, const int *dtypes_host) {
	NGP_REQUIRE(comm && (n_bufs == 0 || (bufs_host && counts_host && dtypes_host)), NGP_E_ARG, "ngp_allreduce_grads: null pointer");
	NgpComm *c = (NgpComm *)comm;
	if (n_bufs == 0) return 0;
	for (int i = 0; i < n_bufs; ++i) NGP_REQUIRE(dtypes_host[i] == NGP_F32 || dtypes_host[i] == NGP_F16, NGP_E_DTYPE, "ngp_allreduce_grads: bad dtype %d", dtypes_host[i]);
	RcclGroup grp;
	for (int i = 0; i < n_bufs; ++i)
		if (counts_host[i]) GROUP_CALL(grp, g_rccl.AllReduce(bufs_host[i], bufs_host[i], (size_t)counts_host[i], nccl_dtype(dtypes_host[i]), 0, c->nccl, (hipStream_t)stream), "ncclAllReduce");
	GROUP_FINISH(grp, "ncclGroupStart/End(ngp_allreduce_grads)");
	return 0;
}

// reduce-scatter of the buckets `first_bucket .. last_bucket` of `grad` (element type `dtype`) according to `plan` - every rank receives the sum of its shard IN PLACE -
// plus, in the same group, the all-reduce of the fp32 tail (when `tail_f32` is given and the last bucket is included) and of `extra_f32[extra_count]` (the MLP gradients)
int ngp_dp_reduce(void *comm, hipStream_t s, const NgpDpPlan *plan, void *grad, int dtype, uint32_t first_bucket, uint32_t last_bucket, float *tail_f32, float *extra_f32, uint64_t extra_count) {
	NgpComm *c = (NgpComm *)comm;
	const size_t es = dtype == NGP_F16 ? 2 : 4;
	RcclGroup grp;
	for (uint32_t b = first_bucket; b <= last_bucket && b < plan->n_buckets; ++b) {
		if (!plan->shard_count[b]) continue;
		char *base = (char *)grad + plan->cut[b] * es;
		GROUP_CALL(grp, g_rccl.ReduceScatter(base, base + (size_t)plan->rank * plan->shard_count[b] * es, (size_t)plan->shard_count[b], nccl_dtype(dtype), 0, c->nccl, s), "ncclReduceScatter");
	}
	if (tail_f32 && plan->tail_count && last_bucket + 1 >= plan->n_buckets)
		GROUP_CALL(grp, g_rccl.AllReduce(tail_f32 + plan->tail_begin, tail_f32 + plan->tail_begin, (size_t)plan->tail_count, 7, 0, c->nccl, s), "ncclAllReduce(tail)");
	if (extra_f32 && extra_count) GROUP_CALL(grp, g_rccl.AllReduce(extra_f32, extra_f32, (size_t)extra_count, 7, 0, c->nccl, s), "ncclAllReduce(mlp)");
	GROUP_FINISH(grp, "ncclGroupStart/End(ngp_dp_reduce)");
	return 0;
}

// all-gather, in place, of every rank's shard of up to four buffers laid out like the table (the updated parameters; at checkpoint time also the Adam moments)
NGP_API int ngp_dp_allgather(void *comm, void *stream, const NgpDpPlan *plan, int n_bufs, void *const *bufs_host, const int *dtypes_host) {
	NGP_REQUIRE(comm && plan && (n_bufs == 0 || (bufs_host && dtypes_host)), NGP_E_ARG, "ngp_dp_allgather: null pointer");
	NgpComm *c = (NgpComm *)comm;
	NGP_REQUIRE(plan->world == c->world && plan->rank == c->rank, NGP_E_ARG, "ngp_dp_allgather: plan is for rank %d of %d, communicator is rank %d of %d", plan->rank, plan->world, c->rank, c->world);
	if (n_bufs == 0) return 0;
	for (int i = 0; i < n_bufs; ++i) NGP_REQUIRE(dtypes_host[i] == NGP_F32 || dtypes_host[i] == NGP_F16, NGP_E_DTYPE, "ngp_dp_allgather: bad dtype %d", dtypes_host[i]);   // (before the group opens: an early return must not leave it open)
	RcclGroup grp;
	for (int i = 0; i < n_bufs; ++i) {
		const size_t es = dtypes_host[i] == NGP_F16 ? 2 : 4;
		for (uint32_t b = 0; b < plan->n_buckets; ++b) {
			if (!plan->shard_count[b]) continue;
			char *base = (char *)bufs_host[i] + plan->cut[b] * es;
			GROUP_CALL(grp, g_rccl.AllGather(base + (size_t)plan->rank * plan->shard_count[b] * es, base, (size_t)plan->shard_count[b], nccl_dtype(dtypes_host[i]), c->nccl, (hipStream_t)stream), "ncclAllGather");
		}
	}
	GROUP_FINISH(grp, "ncclGroupStart/End(ngp_dp_allgather)");
	return 0;
}

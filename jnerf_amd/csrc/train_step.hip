// Host-side driver of one training iteration: the launch sequence of Runner.train's body (runner/runner.py:71-76) issued from native code.
// No kernels here - every stage is one of the library's own entry points, called in the order jnerf_amd/fastpath.py calls them; the point is to
// cross the Python/ctypes boundary once per iteration instead of eleven times (the host had become the pacing side at ~0.45 ms per iteration).
#include "ngp_common.h"
#include "mlp_tail.h"
#include <stdlib.h>
#include <mutex>
#include <utility>
#include <vector>

namespace {
std::mutex g_mu;
std::vector<std::pair<hipEvent_t, hipEvent_t>> g_pending, g_free;
std::pair<hipEvent_t, hipEvent_t> g_boundary{nullptr, nullptr};   // open bracket: first = end of the previous call
struct Bracket {                                       // HIP event pair around one stage, on the stream the stage is launched on
	hipStream_t s; bool on; std::pair<hipEvent_t, hipEvent_t> ev{nullptr, nullptr};
	Bracket(hipStream_t s_, bool on_) : s(s_), on(on_) {
		if (!on) return;
		std::lock_guard<std::mutex> lk(g_mu);
		if (!g_free.empty()) { ev = g_free.back(); g_free.pop_back(); }
		else if (hipEventCreate(&ev.first) != hipSuccess || hipEventCreate(&ev.second) != hipSuccess) { on = false; return; }
		hipEventRecord(ev.first, s);
	}
	~Bracket() {
		if (!on) return;
		hipEventRecord(ev.second, s);
		std::lock_guard<std::mutex> lk(g_mu);
		g_pending.push_back(ev);
	}
};
}  // namespace

NGP_API int ngp_train_step_timings(float *ms_out, int max) {
	NGP_REQUIRE(ms_out || max == 0, NGP_E_ARG, "ngp_train_step_timings: null output");
	std::lock_guard<std::mutex> lk(g_mu);
	int n = 0;
	for (auto &ev : g_pending) {
		if (n < max) {
			hipError_t e = hipEventSynchronize(ev.second);
			float ms = 0.f;
			if (e == hipSuccess) e = hipEventElapsedTime(&ms, ev.first, ev.second);
			if (e != hipSuccess) { ngp_set_error("ngp_train_step_timings: %s", hipGetErrorString(e)); return -(int)e; }
			ms_out[n++] = ms;
		}
		g_free.push_back(ev);
	}
	g_pending.clear();
	return n;
}

// ---- batch hand-over by device flag (r3).  The sampling streams march batches two steps ahead; handing batch i to the training stream through an event costs that
// stream ~29 us per iteration on MI355X (a marker packet on one queue, a barrier packet on the other - profiles/r02_lego_timeline.txt), although the batch has been
// ready for a whole iteration.  Instead the sampling stream ends a batch with k_flag_signal (its own launch: the producing kernels have completed and released their
// writes when it runs) and ngp_train_step starts with k_flag_wait, one wavefront that finds the flag already set in the normal case.  Kernel boundaries on both sides
// provide the release / acquire; the spin is bounded (2 s) and reports through `status` instead of hanging the GPU.
__global__ void k_flag_signal(uint32_t *flag, uint32_t value) {
	if (threadIdx.x == 0) __hip_atomic_store(flag, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__global__ void k_flag_wait(const uint32_t *flag, uint32_t value, uint32_t *status) {
	if (threadIdx.x != 0) return;
	const unsigned long long t0 = wall_clock64();                // constant 100 MHz counter
	while ((int32_t)(__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - value) < 0) {
		__builtin_amdgcn_s_sleep(16);
		if (wall_clock64() - t0 > 200000000ull) { if (status) atomicOr(status, 1u); break; }
	}
}
NGP_API int ngp_flag_signal(void *stream, uint32_t *flag, uint32_t value) {
	NGP_REQUIRE(flag, NGP_E_ARG, "ngp_flag_signal: null flag");
	hipLaunchKernelGGL(k_flag_signal, dim3(1), dim3(64), 0, (hipStream_t)stream, flag, value);      // (not through NGP_LAUNCH: a hand-over is not a kernel of the path - the per-kernel event brackets of csrc/prof.hip skip it)
	NGP_LAUNCH_CHECK("ngp_flag_signal");
	return 0;
}
NGP_API int ngp_flag_wait(void *stream, const uint32_t *flag, uint32_t value, uint32_t *status) {
	NGP_REQUIRE(flag, NGP_E_ARG, "ngp_flag_wait: null flag");
	hipLaunchKernelGGL(k_flag_wait, dim3(1), dim3(64), 0, (hipStream_t)stream, flag, value, status);  // (likewise: its duration is the time it WAITS)
	NGP_LAUNCH_CHECK("ngp_flag_wait");
	return 0;
}

// ---- data parallel, overlapped variant: the library's communication stream and the two markers of one iteration (created on first use, per process)
namespace {
struct DpSide {
	hipStream_t stream = nullptr; hipEvent_t coarse = nullptr, reduced = nullptr; bool ok = false;
	DpSide() {
		ok = hipStreamCreateWithFlags(&stream, hipStreamNonBlocking) == hipSuccess && hipEventCreateWithFlags(&coarse, hipEventDisableTiming) == hipSuccess &&
		     hipEventCreateWithFlags(&reduced, hipEventDisableTiming) == hipSuccess;
	}
};
}  // namespace

// one Adam+EMA sweep over elements [off, off + cnt) of optimiser tensor t (gradient: the fp32 buffer, or the fp16 wire buffer with its scale divided out)
static int sweep_range(void *stream, const NgpTrainStep *a, int t, uint64_t off, uint64_t cnt, bool wire, int zero_grad) {
	if (!cnt) return 0;
	void *g = wire ? (void *)((__half *)a->grad_wire + off) : (void *)(a->g[t] + off);
	return ngp_adam_ema_step_scaled(stream, cnt, a->p[t] + off, g, wire ? NGP_F16 : NGP_F32, a->m[t] + off, a->v[t] + off, a->ema[t] ? a->ema[t] + off : nullptr,
	                                a->p_half[t] ? (void *)((__half *)a->p_half[t] + off) : nullptr, a->lr, a->beta0, a->beta1, a->eps, a->step, a->ema_decay, zero_grad,
	                                wire ? 1.0f / a->wire_scale : 1.0f);
}

NGP_API int ngp_train_step(void *stream, const NgpTrainStep *a) {
	NGP_REQUIRE(a, NGP_E_ARG, "ngp_train_step: null argument block");
	NGP_REQUIRE(a->n_opt >= 0 && a->n_opt <= 4, NGP_E_ARG, "ngp_train_step: n_opt %d out of range", a->n_opt);
	NGP_REQUIRE(a->phase == NGP_PHASE_ALL || a->phase == NGP_PHASE_BACKWARD || a->phase == NGP_PHASE_SWEEP, NGP_E_ARG, "ngp_train_step: bad phase %d", a->phase);
	NGP_REQUIRE(a->dtype == NGP_F16 || a->dtype == NGP_F32, NGP_E_DTYPE, "ngp_train_step: bad dtype %d", a->dtype);
	const bool do_bwd = a->phase != NGP_PHASE_SWEEP, do_sweep = a->phase != NGP_PHASE_BACKWARD && a->run_optimizer;
	const bool dp = a->comm != nullptr;
	if (dp) {
		NGP_REQUIRE(a->phase == NGP_PHASE_ALL && a->run_optimizer, NGP_E_ARG, "ngp_train_step: a communicator needs phase NGP_PHASE_ALL with run_optimizer");
		NGP_REQUIRE(a->dp && a->dp_table >= 0 && a->dp_table < a->n_opt && a->g[a->dp_table] == a->table_grad, NGP_E_ARG, "ngp_train_step: data parallel needs a plan and dp_table naming the hash table among the optimiser tensors");
		NGP_REQUIRE(a->grad_overwrite, NGP_E_ARG, "ngp_train_step: data parallel needs grad_overwrite (the reduce-scatter leaves partial sums outside the rank's shard)");
		NGP_REQUIRE(a->dp->tail_begin + a->dp->tail_count == a->n_params, NGP_E_ARG, "ngp_train_step: plan covers %llu elements, table has %llu", (unsigned long long)(a->dp->tail_begin + a->dp->tail_count), (unsigned long long)a->n_params);
		NGP_REQUIRE(!a->grad_wire || (a->wire_scale > 0.f), NGP_E_ARG, "ngp_train_step: wire_scale must be positive");
	}
	// host-driven exchange (a process group without RCCL): NGP_PHASE_SWEEP with a plan but no communicator sweeps this rank's shard (+ the replicated tail) only - the
	// caller has summed the gradient over the ranks and gathers the updated shards itself.  Same shard arithmetic as the in-library path, which is how two ranks
	// sharing one GPU (tests) exercise it.
	const bool host_sharded = !dp && a->dp != nullptr && a->phase == NGP_PHASE_SWEEP;
	if (host_sharded) {
		NGP_REQUIRE(a->dp_table >= 0 && a->dp_table < a->n_opt && a->g[a->dp_table] == a->table_grad, NGP_E_ARG, "ngp_train_step: a plan needs dp_table naming the hash table among the optimiser tensors");
		NGP_REQUIRE(a->dp->tail_begin + a->dp->tail_count == a->n_params, NGP_E_ARG, "ngp_train_step: plan covers %llu elements, table has %llu", (unsigned long long)(a->dp->tail_begin + a->dp->tail_count), (unsigned long long)a->n_params);
	}
	if (do_bwd) NGP_REQUIRE(a->coords && a->pos && a->numsteps && a->numsteps_compacted && a->bg && a->target && a->rgb && a->loss_grad, NGP_E_ARG, "ngp_train_step: null batch pointer");
	const int lay = NGP_LAYOUT_SOA | NGP_WEIGHTS_PACKED;
	const float *dirs = a->coords + 4;                       // NerfCoordinate = {pos[3], dt, dir[3]}: directions at stride 7
	int rc;
	hipStream_t hs = (hipStream_t)stream;
	static DpSide *side = nullptr;
	const bool overlap = dp && a->dp_overlap && a->dp->n_buckets == 2;
	if (overlap && !side) { side = new DpSide(); }
	NGP_REQUIRE(!overlap || side->ok, NGP_E_ARG, "ngp_train_step: cannot create the communication stream");
	if (a->timed_stage == NGP_STAGE_BOUNDARY && do_bwd) {    // close the bracket opened at the end of the previous call
		std::lock_guard<std::mutex> lk(g_mu);
		if (g_boundary.first) { hipEventRecord(g_boundary.second, hs); g_pending.push_back(g_boundary); g_boundary = {nullptr, nullptr}; }
	}
#define STAGE(id, call) do { Bracket br(hs, a->timed_stage == (id)); rc = (call); } while (0); if (rc) return rc
	const int T = a->dtype, ow = a->grad_overwrite != 0;
	if (do_bwd && a->wait_flag && (rc = ngp_flag_wait(stream, a->wait_flag, a->wait_value, a->wait_status))) return rc;      // the batch's hand-over from the sampling stream
	// the flat fp32 weight pack among the optimiser tensors (fp32 network): its sweep also writes the next iteration's MFMA fragments (ngp_mlp32_sweep_pack)
	int t_pack = -1;
	bool t_pack_swept = false;                                   // (r6) the pack's sweep rode in the hash backward's launches
	int t_table = -1;                                            // (r6) the hash table among the optimiser tensors, when ITS sweep rode in the accumulate kernel
	if (T == NGP_F32) for (int t = 0; t < a->n_opt; ++t)
		if (a->p[t] == (float *)a->wd && a->numel[t] == 10240 && (const float *)a->wc == (const float *)a->wd + 3072 && a->ema[t] == a->p[t] && !a->p_half[t] && a->g[t] == a->wgrad_flat && ow) t_pack = t;
	// fp16 network: the two weight packs (density MLP 3072, colour MLP 7168 elements) whose gradients tile the flat weight-gradient buffer - swept by the slab reduction
	// itself (ngp_reduce_slabs_sweep) when backward and sweep are one call on one GPU (no exchange step between them)
	int t_mlp16[2] = {-1, -1};
	if (T == NGP_F16 && do_bwd && do_sweep && !dp && !host_sharded && ow && a->wgrad_flat) {
		for (int t = 0; t < a->n_opt; ++t) {
			if (a->g[t] == a->wgrad_flat && a->numel[t] == 3072) t_mlp16[0] = t;
			if (a->g[t] == a->wgrad_flat + 3072 && a->numel[t] == 7168) t_mlp16[1] = t;
		}
		if (t_mlp16[0] < 0 || t_mlp16[1] < 0) t_mlp16[0] = t_mlp16[1] = -1;
	}
	if (getenv("NGP_NO_FUSED_MLP_TAIL")) { t_pack = -1; t_mlp16[0] = t_mlp16[1] = -1; }           // A/B hook
	if (do_bwd) {
	// largest |dL/dfeature| per level: written by the field backward kernel's epilogue when the scatter takes the binned path (one pass and one launch less)
	AbsmaxOut am = ngp_hash_bwd_absmax_slots(a->level_table_host, a->n, T, NGP_F32, a->hash_workspace, a->hash_workspace_bytes, a->table_grad);
	if (T == NGP_F16) {
		STAGE(NGP_STAGE_PACK, ngp_field_pack_weights(stream, a->wd, a->wc, a->packed_weights));
		STAGE(NGP_STAGE_HASH_FWD, ngp_hash_encode_fwd(stream, a->n, a->pos, 3, a->table, a->level_table_host, a->feat, NGP_F16, NGP_LAYOUT_SOA, a->n_valid));
		STAGE(NGP_STAGE_FIELD_FWD, ngp_field_fwd(stream, a->n, a->feat, lay, dirs, 7, a->packed_weights, nullptr, a->out, NGP_F16, a->n_valid));
	} else {
		if (!(a->frags_fresh && t_pack >= 0)) {                  // (else: the previous call's fused tail left the fragments of the current weights in packed_weights)
			STAGE(NGP_STAGE_PACK, ngp_field32_pack_weights(stream, (const float *)a->wd, (const float *)a->wc, (float *)a->packed_weights));
		}
		STAGE(NGP_STAGE_HASH_FWD, ngp_hash_encode_fwd(stream, a->n, a->pos, 3, a->table, a->level_table_host, a->feat, NGP_F32, NGP_LAYOUT_SOA, a->n_valid));
		STAGE(NGP_STAGE_FIELD_FWD, ngp_field32_fwd(stream, a->n, (const float *)a->feat, lay, dirs, 7, (const float *)a->packed_weights, nullptr, (float *)a->out, a->n_valid));
	}
	// compositing forward + Huber + compositing backward: one launch (r5; bit-identical to the two launches it replaces, which remain the module path's)
	const bool split_composite = getenv("NGP_SPLIT_COMPOSITE") != nullptr;            // A/B hook (read per call: the tests toggle it)
	if (!split_composite) {
		STAGE(NGP_STAGE_COMPOSITE_FWD, ngp_composite_train(stream, a->n_rays, a->n, a->out, T, a->coords, a->numsteps, a->numsteps_compacted, a->bg, a->cascades, a->rgb,
		                                                   a->target, a->huber_delta, a->loss, a->loss_grad, a->density_grid_mean, a->dout));
	} else {
		STAGE(NGP_STAGE_COMPOSITE_FWD, ngp_composite_fwd_huber(stream, a->n_rays, a->out, T, a->coords, a->numsteps, a->numsteps_compacted, a->bg, a->cascades, a->rgb,
		                                                       a->target, a->huber_delta, a->loss, a->loss_grad));
		STAGE(NGP_STAGE_COMPOSITE_BWD, ngp_composite_bwd(stream, a->n_rays, a->n, a->out, T, a->coords, a->numsteps_compacted, a->loss_grad, a->rgb, a->density_grid_mean, a->cascades, a->dout, 0));
	}
	if (T == NGP_F16) {
		STAGE(NGP_STAGE_FIELD_BWD, ngp_field_bwd_am(stream, a->n, a->feat, lay, dirs, 7, a->packed_weights, nullptr, a->dout, NGP_F16, a->dfeat, a->wgrad_slabs, a->n_slabs, a->n_valid, &am));
	} else {
		STAGE(NGP_STAGE_FIELD_BWD, ngp_field32_bwd_am(stream, a->n, (const float *)a->feat, lay, dirs, 7, (const float *)a->packed_weights, nullptr, (const float *)a->dout, (float *)a->dfeat,
		                                               a->wgrad_slabs, a->n_slabs, a->n_valid, &am));
	}
	// (r6) fp32 configuration, single GPU, backward and sweep in one call: the slab reduction and the pack's sweep + fragment packing RIDE in the hash backward's two record
	// launches (mlp_tail.h) instead of being two small launches in front of / behind it - when the scatter takes the path that has both (it says so: tail_taken)
	TailJobs tail = no_tail_jobs();
	int tail_taken = 0;
	const bool want_tail = T == NGP_F32 && t_pack >= 0 && do_sweep && !dp && !host_sharded && a->wgrad_flat && !getenv("NGP_NO_TAIL_RIDE");
	if (want_tail) {
		tail.slabs = a->wgrad_slabs; tail.n_slabs = a->n_slabs; tail.width = 10240u; tail.reduce_out = a->wgrad_flat;
		tail.pack = a->p[t_pack]; tail.m = a->m[t_pack]; tail.v = a->v[t_pack]; tail.packed_out = (float *)a->packed_weights;
		tail.c = adam_consts(a->lr, a->beta0, a->beta1, a->eps, a->step, a->ema_decay, 1.0f);
		tail.pack_table = ngp_mlp32_pack_table(stream);
		tail.do_reduce = 1; tail.do_sweep = 1;
	}
	// fp16 configuration, single GPU: the slab reduction also sweeps the two weight packs (their gradient is the sum it has just formed) - and rides the same way
	const float *pk16[2][5]; uint32_t begin16[2], count16[2];
	const bool want_tail16 = t_mlp16[0] >= 0 && !getenv("NGP_NO_TAIL_RIDE");
	if (t_mlp16[0] >= 0) {
		for (int k = 0; k < 2; ++k) {
			const int t = t_mlp16[k];
			pk16[k][0] = a->p[t]; pk16[k][1] = a->m[t]; pk16[k][2] = a->v[t]; pk16[k][3] = a->ema[t]; pk16[k][4] = (const float *)a->p_half[t];
			begin16[k] = (uint32_t)(a->g[t] - a->wgrad_flat); count16[k] = (uint32_t)a->numel[t];
		}
	}
	if (want_tail16) {
		tail.slabs = a->wgrad_slabs; tail.n_slabs = a->n_slabs; tail.width = 10240u; tail.reduce_out = a->wgrad_flat;
		for (int k = 0; k < 2; ++k) {
			PackSweep &w = k ? tail.b16 : tail.a16;
			w = PackSweep{(float *)pk16[k][0], (float *)pk16[k][1], (float *)pk16[k][2], (float *)pk16[k][3], (__half *)pk16[k][4], begin16[k], count16[k]};
		}
		tail.c = adam_consts(a->lr, a->beta0, a->beta1, a->eps, a->step, a->ema_decay, 1.0f);
		tail.do_reduce = 1; tail.do_sweep16 = 1;
	}
	if (want_tail || want_tail16) {
		// (the reduction follows the scatter call below if that call did not carry it)
	} else if (t_mlp16[0] >= 0) {
		STAGE(NGP_STAGE_REDUCE_SLABS, ngp_reduce_slabs_sweep(stream, a->wgrad_slabs, a->n_slabs, 10240, a->wgrad_flat, pk16, begin16, count16, a->lr, a->beta0, a->beta1, a->eps, a->step, a->ema_decay));
	} else {
		STAGE(NGP_STAGE_REDUCE_SLABS, ngp_reduce_slabs(stream, a->wgrad_slabs, a->n_slabs, 10240, a->wgrad_flat, ow ? 0 : 1));
	}
	// (r6) single GPU, backward and sweep in one call: the TABLE's sweep rides in the accumulate kernel(s), which apply Adam + EMA to each entry in place of storing its
	// gradient (k_bin_accumulate2_adam | k_bin_accumulate_adam_*; the call says whether its path could: adam_taken) - table_grad is not written in such a call
	AdamRide ride{nullptr, nullptr, nullptr, nullptr, AdamConsts{}, 0};
	if (do_sweep && !dp && !host_sharded && ow && !getenv("NGP_NO_ADAM_RIDE")) {
		// the hash table among the optimiser tensors: fp32 configuration - the master IS what the gathers read; fp16 configuration - they read its fp16 shadow
		for (int t = 0; t < a->n_opt; ++t)
			if (a->g[t] == a->table_grad && a->numel[t] == a->n_params && (!a->ema[t] || a->ema[t] == a->p[t]) &&
			    (T == NGP_F32 ? (a->p[t] == (float *)a->table && !a->p_half[t]) : (a->p_half[t] == a->table))) t_table = t;
		if (t_table >= 0) {
			ride.p = a->p[t_table]; ride.m = a->m[t_table]; ride.v = a->v[t_table]; ride.p_half = (__half *)a->p_half[t_table]; ride.ema = a->ema[t_table] ? 1 : 0;
			ride.c = adam_consts(a->lr, a->beta0, a->beta1, a->eps, a->step, a->ema_decay, 1.0f);
		}
	}
	int adam_taken = 0;
	STAGE(NGP_STAGE_HASH_BWD, ngp_hash_encode_bwd_ws_marked(stream, a->n, a->pos, 3, a->dfeat, a->level_table_host, a->table_grad, a->n_params, T, NGP_F32, NGP_LAYOUT_SOA, ow ? 1 : 0, a->n_valid,
	                                                        a->hash_workspace, a->hash_workspace_bytes, overlap ? side->coarse : nullptr, am.parts != nullptr, (want_tail || want_tail16) ? &tail : nullptr, &tail_taken,
	                                                        t_table >= 0 ? &ride : nullptr, &adam_taken));
	if (!adam_taken) t_table = -1;
	if (want_tail && !tail_taken) { STAGE(NGP_STAGE_REDUCE_SLABS, ngp_reduce_slabs(stream, a->wgrad_slabs, a->n_slabs, 10240, a->wgrad_flat, 0)); }
	if (want_tail16 && !tail_taken) {
		STAGE(NGP_STAGE_REDUCE_SLABS, ngp_reduce_slabs_sweep(stream, a->wgrad_slabs, a->n_slabs, 10240, a->wgrad_flat, pk16, begin16, count16, a->lr, a->beta0, a->beta1, a->eps, a->step, a->ema_decay));
	}
	if (tail_taken && want_tail) t_pack_swept = true;
	}
	// ---- exchange step (data parallel): every rank ends up with the summed gradient of its shard of the table, of the tail and of the MLP pack
	const bool wire = dp && a->grad_wire != nullptr;
	if (dp) {
		const NgpDpPlan *pl = a->dp;
		void *gbuf = wire ? a->grad_wire : (void *)a->table_grad;
		const int gdt = wire ? NGP_F16 : NGP_F32;
		auto to_wire = [&](hipStream_t st, uint32_t b) -> int {      // fp32 -> scaled fp16, bucket b only
			const uint64_t off = pl->cut[b], cnt = pl->cut[b + 1] - pl->cut[b];
			return ngp_grad_to_half_scaled(st, cnt, a->table_grad + off, (__half *)a->grad_wire + off, 0, a->wire_scale);
		};
		if (overlap) {
			// bucket 0 (coarse levels) + the MLP gradients on the communication stream, as soon as the coarse accumulate launch has finished - under the fine levels' accumulate
			hipStreamWaitEvent(side->stream, side->coarse, 0);
			if (wire && (rc = to_wire(side->stream, 0))) return rc;
			if ((rc = ngp_dp_reduce(a->comm, side->stream, pl, gbuf, gdt, 0, 0, nullptr, a->wgrad_flat, 10240))) return rc;
			hipEventRecord(side->reduced, side->stream);
			if (wire && (rc = to_wire(hs, 1))) return rc;
			if ((rc = ngp_dp_reduce(a->comm, hs, pl, gbuf, gdt, 1, 1, a->table_grad, nullptr, 0))) return rc;
			hipStreamWaitEvent(hs, side->reduced, 0);
		} else {
			if (wire) for (uint32_t b = 0; b < pl->n_buckets; ++b) if ((rc = to_wire(hs, b))) return rc;
			if ((rc = ngp_dp_reduce(a->comm, hs, pl, gbuf, gdt, 0, pl->n_buckets - 1, a->table_grad, a->wgrad_flat, 10240))) return rc;
		}
	}
	if (do_sweep) {
		int largest = 0;
		for (int t = 1; t < a->n_opt; ++t) if (a->numel[t] > a->numel[largest]) largest = t;
		for (int t = 0; t < a->n_opt; ++t) {
			Bracket br(hs, a->timed_stage == NGP_STAGE_ADAM && t == largest);
			if ((dp || host_sharded) && t == a->dp_table) {  // this rank's shards (1/N of the 28 B/parameter stream) + the replicated tail
				const NgpDpPlan *pl = a->dp;
				for (uint32_t b = 0; b < pl->n_buckets; ++b) if ((rc = sweep_range(stream, a, t, pl->shard_begin[b], pl->shard_count[b], wire, 0))) return rc;
				if ((rc = sweep_range(stream, a, t, pl->tail_begin, pl->tail_count, false, 0))) return rc;
			} else if (t == t_table) {
				continue;                                        // swept by the accumulate kernel(s) of the hash backward
			} else if (t == t_mlp16[0] || t == t_mlp16[1]) {
				continue;                                        // already swept by ngp_reduce_slabs_sweep
			} else if (t == t_pack) {
				if (t_pack_swept) continue;
				if ((rc = ngp_mlp32_sweep_pack(stream, a->p[t], a->g[t], a->m[t], a->v[t], a->lr, a->beta0, a->beta1, a->eps, a->step, a->ema_decay, (float *)a->packed_weights))) return rc;
			} else if ((rc = ngp_adam_ema_step(stream, a->numel[t], a->p[t], a->g[t], NGP_F32, a->m[t], a->v[t], a->ema[t], a->p_half[t], a->lr, a->beta0, a->beta1, a->eps,
			                                   a->step, a->ema_decay, ow ? 0 : 1))) return rc;
		}
	}
	if (dp) {                                                    // everyone gets everyone's updated shard of what the kernels read
		const int t = a->dp_table;
		void *bufs[2]; int dts[2]; int nb = 0;
		if (a->dp_gather_master || !a->p_half[t]) { bufs[nb] = a->p[t]; dts[nb++] = NGP_F32; }
		if (a->p_half[t]) { bufs[nb] = a->p_half[t]; dts[nb++] = NGP_F16; }
		if ((rc = ngp_dp_allgather(a->comm, stream, a->dp, nb, bufs, dts))) return rc;
	}
#undef STAGE
	if (a->timed_stage == NGP_STAGE_BOUNDARY && a->phase != NGP_PHASE_BACKWARD) {
		std::lock_guard<std::mutex> lk(g_mu);
		if (!g_free.empty()) { g_boundary = g_free.back(); g_free.pop_back(); }
		else if (hipEventCreate(&g_boundary.first) != hipSuccess || hipEventCreate(&g_boundary.second) != hipSuccess) g_boundary = {nullptr, nullptr};
		if (g_boundary.first) hipEventRecord(g_boundary.first, hs);
	}
	return 0;
}


// ---------------------------------------------------------------------------------------------------------------- one inference chunk (Runner.render_img's loop body)
__global__ void k_render_totals(const uint32_t *__restrict__ counters, uint32_t max_samples, unsigned long long *__restrict__ totals) {
	if (threadIdx.x == 0 && blockIdx.x == 0) {                // counters: {-, requested samples, -, samples written}; atomics: chunks may run on two streams at once
		atomicAdd(&totals[0], (unsigned long long)counters[3]);
		if (counters[1] > max_samples) atomicAdd(&totals[1], 1ull);
	}
}

NGP_API int ngp_render_chunk(void *stream, const NgpRenderChunk *a) {
	NGP_REQUIRE(a, NGP_E_ARG, "ngp_render_chunk: null argument block");
	NGP_REQUIRE(a->dtype == NGP_F16 || a->dtype == NGP_F32, NGP_E_DTYPE, "ngp_render_chunk: bad dtype %d", a->dtype);
	NGP_REQUIRE(a->rays_o && a->rays_d && a->bitfield && a->rng_state_host && a->coords && a->pos && a->numsteps && a->numsteps_compacted && a->counters && a->scratch &&
	            a->table && a->level_table_host && a->packed_weights && a->feat && a->out && a->rgb_out && a->alpha_out && a->totals, NGP_E_ARG, "ngp_render_chunk: null pointer");
	if (a->n_rays == 0) return 0;
	int rc;
	const int lay = NGP_LAYOUT_SOA | NGP_WEIGHTS_PACKED;
	if ((rc = ngp_march_rays_compacted_bounds(stream, a->n_rays, a->rays_o, a->rays_d, a->bitfield, a->aabb0, a->aabb1, a->near_distance, a->cone_angle, a->const_dt, a->cascades,
	                                          a->rng_state_host, a->max_samples, a->cap, a->coords, a->numsteps, a->numsteps_compacted, a->counters, a->scratch, a->pos, a->occ_bounds))) return rc;
	const uint32_t *n_valid = a->counters + 3;
	if ((rc = ngp_hash_encode_fwd(stream, a->cap, a->pos, 3, a->table, a->level_table_host, a->feat, a->dtype, NGP_LAYOUT_SOA, n_valid))) return rc;
	if (a->dtype == NGP_F16) rc = ngp_field_fwd(stream, a->cap, a->feat, lay, a->coords + 4, 7, a->packed_weights, nullptr, a->out, NGP_F16, n_valid);
	else rc = ngp_field32_fwd(stream, a->cap, (const float *)a->feat, lay, a->coords + 4, 7, (const float *)a->packed_weights, nullptr, (float *)a->out, n_valid);
	if (rc) return rc;
	if ((rc = ngp_composite_inference(stream, a->n_rays, a->out, a->dtype, a->coords, a->numsteps_compacted, a->cascades, a->rgb_out, a->alpha_out))) return rc;
	NGP_LAUNCH(k_render_totals, dim3(1), dim3(64), 0, (hipStream_t)stream, (const uint32_t *)a->counters, a->max_samples, (unsigned long long *)a->totals);
	NGP_LAUNCH_CHECK("ngp_render_chunk");
	return 0;
}

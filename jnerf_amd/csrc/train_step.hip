// Host-side driver of one training iteration: the launch sequence of Runner.train's body (runner/runner.py:71-76) issued from native code.
// No kernels here - every stage is one of the library's own entry points, called in the order jnerf_amd/fastpath.py calls them; the point is to
// cross the Python/ctypes boundary once per iteration instead of eleven times (the host had become the pacing side at ~0.45 ms per iteration).
#include "ngp_common.h"
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

NGP_API int ngp_train_step(void *stream, const NgpTrainStep *a) {
	NGP_REQUIRE(a, NGP_E_ARG, "ngp_train_step: null argument block");
	NGP_REQUIRE(a->n_opt >= 0 && a->n_opt <= 4, NGP_E_ARG, "ngp_train_step: n_opt %d out of range", a->n_opt);
	NGP_REQUIRE(a->coords && a->pos && a->numsteps && a->numsteps_compacted && a->bg && a->target && a->rgb && a->loss_grad, NGP_E_ARG, "ngp_train_step: null batch pointer");
	const int lay = NGP_LAYOUT_SOA | NGP_WEIGHTS_PACKED;
	const float *dirs = a->coords + 4;                       // NerfCoordinate = {pos[3], dt, dir[3]}: directions at stride 7
	int rc;
	hipStream_t hs = (hipStream_t)stream;
	if (a->timed_stage == NGP_STAGE_BOUNDARY) {              // close the bracket opened at the end of the previous call
		std::lock_guard<std::mutex> lk(g_mu);
		if (g_boundary.first) { hipEventRecord(g_boundary.second, hs); g_pending.push_back(g_boundary); g_boundary = {nullptr, nullptr}; }
	}
#define STAGE(id, call) do { Bracket br(hs, a->timed_stage == (id)); rc = (call); } while (0); if (rc) return rc
	NGP_REQUIRE(a->dtype == NGP_F16 || a->dtype == NGP_F32, NGP_E_DTYPE, "ngp_train_step: bad dtype %d", a->dtype);
	const int T = a->dtype, ow = a->grad_overwrite != 0;
	if (T == NGP_F16) {
		STAGE(NGP_STAGE_PACK, ngp_field_pack_weights(stream, a->wd, a->wc, a->packed_weights));
		STAGE(NGP_STAGE_HASH_FWD, ngp_hash_encode_fwd(stream, a->n, a->pos, 3, a->table, a->level_table_host, a->feat, NGP_F16, NGP_LAYOUT_SOA, a->n_valid));
		STAGE(NGP_STAGE_FIELD_FWD, ngp_field_fwd(stream, a->n, a->feat, lay, dirs, 7, a->packed_weights, nullptr, a->out, NGP_F16, a->n_valid));
	} else {
		STAGE(NGP_STAGE_PACK, ngp_field32_pack_weights(stream, (const float *)a->wd, (const float *)a->wc, (float *)a->packed_weights));
		STAGE(NGP_STAGE_HASH_FWD, ngp_hash_encode_fwd(stream, a->n, a->pos, 3, a->table, a->level_table_host, a->feat, NGP_F32, NGP_LAYOUT_SOA, a->n_valid));
		STAGE(NGP_STAGE_FIELD_FWD, ngp_field32_fwd(stream, a->n, (const float *)a->feat, lay, dirs, 7, (const float *)a->packed_weights, nullptr, (float *)a->out, a->n_valid));
	}
	STAGE(NGP_STAGE_COMPOSITE_FWD, ngp_composite_fwd_huber(stream, a->n_rays, a->out, T, a->coords, a->numsteps, a->numsteps_compacted, a->bg, a->cascades, a->rgb,
	                                                       a->target, a->huber_delta, a->loss, a->loss_grad));
	STAGE(NGP_STAGE_COMPOSITE_BWD, ngp_composite_bwd(stream, a->n_rays, a->n, a->out, T, a->coords, a->numsteps_compacted, a->loss_grad, a->rgb, a->density_grid_mean, a->cascades, a->dout, 0));
	if (T == NGP_F16) {
		STAGE(NGP_STAGE_FIELD_BWD, ngp_field_bwd(stream, a->n, a->feat, lay, dirs, 7, a->packed_weights, nullptr, a->dout, NGP_F16, a->dfeat, a->wgrad_slabs, a->n_slabs, a->n_valid));
	} else {
		STAGE(NGP_STAGE_FIELD_BWD, ngp_field32_bwd(stream, a->n, (const float *)a->feat, lay, dirs, 7, (const float *)a->packed_weights, nullptr, (const float *)a->dout, (float *)a->dfeat,
		                                            a->wgrad_slabs, a->n_slabs, a->n_valid));
	}
	STAGE(NGP_STAGE_REDUCE_SLABS, ngp_reduce_slabs(stream, a->wgrad_slabs, a->n_slabs, 10240, a->wgrad_flat, ow ? 0 : 1));
	STAGE(NGP_STAGE_HASH_BWD, ngp_hash_encode_bwd_ws(stream, a->n, a->pos, 3, a->dfeat, a->level_table_host, a->table_grad, a->n_params, T, NGP_F32, NGP_LAYOUT_SOA, ow ? 1 : 0, a->n_valid,
	                                                 nullptr, a->hash_workspace, a->hash_workspace_bytes));
	if (a->run_optimizer) {
		int largest = 0;
		for (int t = 1; t < a->n_opt; ++t) if (a->numel[t] > a->numel[largest]) largest = t;
		for (int t = 0; t < a->n_opt; ++t) {
			Bracket br(hs, a->timed_stage == NGP_STAGE_ADAM && t == largest);
			if ((rc = ngp_adam_ema_step(stream, a->numel[t], a->p[t], a->g[t], NGP_F32, a->m[t], a->v[t], a->ema[t], a->p_half[t], a->lr, a->beta0, a->beta1, a->eps,
			                            a->step, a->ema_decay, ow ? 0 : 1))) return rc;
		}
	}
#undef STAGE
	if (a->timed_stage == NGP_STAGE_BOUNDARY) {
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
	if ((rc = ngp_march_rays_compacted_pos(stream, a->n_rays, a->rays_o, a->rays_d, a->bitfield, a->aabb0, a->aabb1, a->near_distance, a->cone_angle, a->const_dt, a->cascades,
	                                       a->rng_state_host, a->max_samples, a->cap, a->coords, a->numsteps, a->numsteps_compacted, a->counters, a->scratch, a->pos))) return rc;
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

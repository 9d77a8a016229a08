// Per-kernel HIP-event brackets for bench.py's roofline object (measurement plumbing, no arithmetic).
// Every launch in the library goes through NGP_LAUNCH (ngp_common.h), which registers the kernel's base name once per call site and - only while that
// kernel is enabled with ngp_prof_enable - records an event pair around the launch ON THE STREAM THE KERNEL IS LAUNCHED ON (the marcher runs on the
// Runner's side streams, the dense-level scatter on the library's helper stream: torch.cuda.Event on the current stream would not see them).
// An event pair costs ~5 us of GPU time per launch, so bench.py enables everything during its probe steps and only the dominant kernel in the timed region.
#include "ngp_common.h"
#include <mutex>
#include <string>
#include <utility>
#include <vector>
#include <string.h>

namespace {
struct Kernel { std::string name; bool enabled = false; std::vector<std::pair<hipEvent_t, hipEvent_t>> pending; };
std::mutex g_mu;
std::vector<Kernel> g_kernels;
std::vector<std::pair<hipEvent_t, hipEvent_t>> g_free;
std::string g_enable_spec;          // "" = nothing, "*" = everything, else comma-separated base names
bool spec_has(const std::string &name) {
	if (g_enable_spec == "*") return true;
	size_t pos = 0;
	while (pos <= g_enable_spec.size()) {
		size_t e = g_enable_spec.find(',', pos); if (e == std::string::npos) e = g_enable_spec.size();
		if (g_enable_spec.compare(pos, e - pos, name) == 0 && e - pos == name.size()) return true;
		pos = e + 1;
	}
	return false;
}
}  // namespace

int g_ngp_prof_on = 0;

int ngp_prof_register(const char *expr) {
	// "(k_hash_fwd<T, L>)" -> "k_hash_fwd"
	std::string s(expr);
	size_t b = s.find_first_not_of("( ");
	size_t e = s.find_first_of("<) ", b == std::string::npos ? 0 : b);
	std::string name = s.substr(b == std::string::npos ? 0 : b, e == std::string::npos ? std::string::npos : e - b);
	std::lock_guard<std::mutex> lk(g_mu);
	for (size_t i = 0; i < g_kernels.size(); ++i) if (g_kernels[i].name == name) return (int)i;
	g_kernels.emplace_back();
	g_kernels.back().name = name;
	g_kernels.back().enabled = spec_has(name);
	return (int)g_kernels.size() - 1;
}

NgpProfScope::NgpProfScope(int id_, hipStream_t s_) : id(id_), s(s_), a(nullptr), b(nullptr) {
	if (!g_ngp_prof_on) return;
	std::lock_guard<std::mutex> lk(g_mu);
	if (!g_kernels[id].enabled) return;
	if (!g_free.empty()) { a = g_free.back().first; b = g_free.back().second; g_free.pop_back(); }
	else if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) { a = b = nullptr; return; }
	hipEventRecord(a, s);
}
NgpProfScope::~NgpProfScope() {
	if (!a) return;
	hipEventRecord(b, s);
	std::lock_guard<std::mutex> lk(g_mu);
	g_kernels[id].pending.emplace_back(a, b);
}

NGP_API int ngp_prof_enable(const char *names) {
	std::lock_guard<std::mutex> lk(g_mu);
	g_enable_spec = names ? names : "";
	for (auto &k : g_kernels) k.enabled = spec_has(k.name);
	g_ngp_prof_on = g_enable_spec.empty() ? 0 : 1;
	return 0;
}

NGP_API int ngp_prof_read(int index, char *name_out, int name_cap, float *ms_out, int max) {
	std::vector<std::pair<hipEvent_t, hipEvent_t>> evs;
	{
		std::lock_guard<std::mutex> lk(g_mu);
		if (index < 0 || index >= (int)g_kernels.size()) return -1;
		if (name_out && name_cap > 0) { strncpy(name_out, g_kernels[index].name.c_str(), (size_t)name_cap - 1); name_out[name_cap - 1] = 0; }
		evs.swap(g_kernels[index].pending);
	}
	int n = 0;
	for (auto &ev : evs) {
		if (n < max && ms_out) {
			float ms = 0.f;
			if (hipEventSynchronize(ev.second) == hipSuccess && hipEventElapsedTime(&ms, ev.first, ev.second) == hipSuccess) ms_out[n++] = ms;
		}
	}
	std::lock_guard<std::mutex> lk(g_mu);
	for (auto &ev : evs) g_free.push_back(ev);
	return n;
}

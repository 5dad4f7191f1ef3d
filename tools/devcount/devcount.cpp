// devcount — device-wide hardware counters around a region of a RUNNING process (rocprofiler-sdk "device counting service").
// Why: `rocprofv3 --pmc` counts per dispatch and SERIALISES the kernels to do so (profiles/r06_pmc_serialises.txt: four streams in flight
// collapse to one kernel at a time with ~120 us gaps), so it cannot see the regime bench.py's `value` is measured in (four batches in
// flight).  This tool library counts on the whole device while the streams run as they do in the bench.
// Use:  ROCP_TOOL_LIBRARIES=$PWD/tools/devcount/libdevcount.so python tools/inflight_counters.py
//       (the python side dlopens the same library and calls devcount_start("A,B,C") / devcount_stop(buf, cap) around its timed region)
// Developer instrument only: nothing in the product links or loads it.
#include <rocprofiler-sdk/registration.h>
#include <rocprofiler-sdk/rocprofiler.h>

#include <cstdio>
#include <cstring>
#include <map>
#include <sstream>
#include <string>
#include <vector>

namespace {
rocprofiler_context_id_t g_ctx{};
rocprofiler_agent_id_t g_agent{};
rocprofiler_counter_config_id_t g_profile{.handle = 0};
bool g_ready = false;
std::map<std::string, rocprofiler_counter_id_t> g_by_name;
std::map<uint64_t, std::string> g_by_id;
std::map<uint64_t, size_t> g_inst;
std::string g_err;

#define DC_CALL(x)                                                                                   \
    do {                                                                                             \
        rocprofiler_status_t s_ = (x);                                                               \
        if (s_ != ROCPROFILER_STATUS_SUCCESS) {                                                      \
            g_err = std::string(#x) + " -> " + rocprofiler_get_status_string(s_);                    \
            fprintf(stderr, "devcount: %s\n", g_err.c_str());                                        \
            return -1;                                                                               \
        }                                                                                            \
    } while (0)

int load_counters() {
    std::vector<rocprofiler_counter_id_t> ids;
    DC_CALL(rocprofiler_iterate_agent_supported_counters(
        g_agent,
        [](rocprofiler_agent_id_t, rocprofiler_counter_id_t *c, size_t n, void *u) {
            auto *v = static_cast<std::vector<rocprofiler_counter_id_t> *>(u);
            for (size_t i = 0; i < n; ++i) v->push_back(c[i]);
            return ROCPROFILER_STATUS_SUCCESS;
        },
        &ids));
    for (auto &c : ids) {
        rocprofiler_counter_info_v1_t info;
        if (rocprofiler_query_counter_info(c, ROCPROFILER_COUNTER_INFO_VERSION_1, &info) != ROCPROFILER_STATUS_SUCCESS) continue;
        g_by_name[info.name] = c;
        g_by_id[c.handle] = info.name;
        g_inst[c.handle] = info.dimensions_instances_count;
    }
    return 0;
}

int tool_init(rocprofiler_client_finalize_t, void *) {
    // first GPU agent
    std::vector<rocprofiler_agent_v0_t> agents;
    rocprofiler_query_available_agents_cb_t cb = [](rocprofiler_agent_version_t, const void **arr, size_t n, void *u) {
        auto *v = static_cast<std::vector<rocprofiler_agent_v0_t> *>(u);
        for (size_t i = 0; i < n; ++i) {
            const auto *a = static_cast<const rocprofiler_agent_v0_t *>(arr[i]);
            if (a->type == ROCPROFILER_AGENT_TYPE_GPU) v->push_back(*a);
        }
        return ROCPROFILER_STATUS_SUCCESS;
    };
    DC_CALL(rocprofiler_query_available_agents(ROCPROFILER_AGENT_INFO_VERSION_0, cb, sizeof(rocprofiler_agent_t), &agents));
    if (agents.empty()) {
        fprintf(stderr, "devcount: no GPU agent\n");
        return -1;
    }
    g_agent = agents[0].id;
    DC_CALL(rocprofiler_create_context(&g_ctx));
    DC_CALL(rocprofiler_configure_device_counting_service(
        g_ctx, rocprofiler_buffer_id_t{.handle = 0}, g_agent,
        [](rocprofiler_context_id_t ctx, rocprofiler_agent_id_t, rocprofiler_device_counting_agent_cb_t set, void *) {
            if (g_profile.handle) set(ctx, g_profile);
        },
        nullptr));
    g_ready = true;
    return 0;
}
void tool_fini(void *) {}
}  // namespace

extern "C" {
__attribute__((visibility("default"))) int devcount_ready() { return g_ready ? 1 : 0; }

// counters: comma separated names.  Starts counting on the whole device.
__attribute__((visibility("default"))) int devcount_start(const char *counters) {
    if (!g_ready) return -2;
    if (g_by_name.empty() && load_counters()) return -1;
    std::vector<rocprofiler_counter_id_t> ids;
    std::stringstream ss(counters);
    std::string nm;
    while (std::getline(ss, nm, ',')) {
        auto it = g_by_name.find(nm);
        if (it == g_by_name.end()) {
            fprintf(stderr, "devcount: unknown counter %s\n", nm.c_str());
            return -3;
        }
        ids.push_back(it->second);
    }
    g_profile.handle = 0;
    DC_CALL(rocprofiler_create_counter_config(g_agent, ids.data(), ids.size(), &g_profile));
    DC_CALL(rocprofiler_start_context(g_ctx));
    return 0;
}

// Reads the counters (accumulated since devcount_start), stops counting, writes one JSON object {name: {sum, n, max}} into out.
__attribute__((visibility("default"))) int devcount_stop(char *out, int cap) {
    if (!g_ready) return -2;
    std::vector<rocprofiler_counter_record_t> rec(1 << 16);
    size_t n = rec.size();
    rocprofiler_status_t s = rocprofiler_sample_device_counting_service(g_ctx, {}, ROCPROFILER_COUNTER_FLAG_NONE, rec.data(), &n);
    rocprofiler_stop_context(g_ctx);
    if (s != ROCPROFILER_STATUS_SUCCESS) {
        fprintf(stderr, "devcount: sample -> %s\n", rocprofiler_get_status_string(s));
        return -1;
    }
    struct agg { double sum = 0, mx = 0; size_t n = 0; };
    std::map<std::string, agg> res;
    for (size_t i = 0; i < n; ++i) {
        rocprofiler_counter_id_t cid{.handle = 0};
        rocprofiler_query_record_counter_id(rec[i].id, &cid);
        auto it = g_by_id.find(cid.handle);
        auto &a = res[it == g_by_id.end() ? std::string("?") : it->second];
        a.sum += rec[i].counter_value;
        a.mx = rec[i].counter_value > a.mx ? rec[i].counter_value : a.mx;
        ++a.n;
    }
    std::string js = "{";
    bool first = true;
    for (auto &kv : res) {
        char b[256];
        snprintf(b, sizeof b, "%s\"%s\": {\"sum\": %.0f, \"n\": %zu, \"max\": %.0f}", first ? "" : ", ", kv.first.c_str(), kv.second.sum, kv.second.n, kv.second.mx);
        js += b;
        first = false;
    }
    js += "}";
    if ((int)js.size() + 1 > cap) return -4;
    memcpy(out, js.c_str(), js.size() + 1);
    return (int)js.size();
}

__attribute__((visibility("default"))) rocprofiler_tool_configure_result_t *rocprofiler_configure(uint32_t, const char *, uint32_t, rocprofiler_client_id_t *id) {
    id->name = "yk_devcount";
    static auto cfg = rocprofiler_tool_configure_result_t{sizeof(rocprofiler_tool_configure_result_t), &tool_init, &tool_fini, nullptr};
    return &cfg;
}
}

// Error reporting, the shared zero page and dropout key/threshold helpers.
#include "kzv_host.h"
#include <cstdlib>
#include "../../include/kzv.h"
#include <cstdarg>
#include <cstdio>
#include <cmath>
#include <mutex>
#include <vector>

static thread_local char g_err[512] = "";

int kzv_fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

int kzv_check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return kzv_fail(KZV_E_HIP, "%s: %s", what, hipGetErrorString(e));
    return KZV_OK;
}

const void* kzv_zero_page() {
    static void* page = nullptr;
    static std::once_flag once;
    std::call_once(once, [] {
        void* p = nullptr;
        if (hipMalloc(&p, 4096) == hipSuccess && hipMemset(p, 0, 4096) == hipSuccess) page = p;
    });
    return page;
}

void kzv_drop_params(float p, unsigned* thr16, float* inv_keep) {
    if (!(p > 0.f)) { *thr16 = 0; *inv_keep = 1.f; return; }
    unsigned t = (unsigned)lrintf(p * 65536.f);
    if (t < 1) t = 1;
    if (t > 65535) t = 65535;
    *thr16 = t;
    *inv_keep = 65536.f / (float)(65536u - t);   // 1 / P(keep) for the threshold actually used
}

extern "C" uint32_t kzv_drop_key(uint64_t seed, uint32_t site) {
    uint64_t x = seed * 0x9E3779B97F4A7C15ull + (uint64_t)site * 0xD1B54A32D192ED03ull + 0x632BE59BD9B4E019ull;
    x ^= x >> 32; x *= 0xD6E8FEB86659FD93ull; x ^= x >> 32; x *= 0xD6E8FEB86659FD93ull; x ^= x >> 32;
    return (uint32_t)x;
}

extern "C" const char* kzv_last_error(void) { return g_err; }

// CUs the GEMM launchers leave to concurrently running collectives (see include/kzv.h)
static int g_cu_reserve = -1;
int kzv_cu_reserve() {
    if (g_cu_reserve < 0) { const char* e = getenv("KZV_CU_RESERVE"); g_cu_reserve = (e && e[0]) ? atoi(e) : 0; }
    return g_cu_reserve;
}
extern "C" int kzv_set_cu_reserve(int n) {
    if (n < 0 || n > 128) return kzv_fail(KZV_E_ARG, "set_cu_reserve: 0..128");
    g_cu_reserve = n;
    return KZV_OK;
}
extern "C" int kzv_version(void) { return 1; }

// ------------------------------------------------------------------------------------------ profiling
namespace {
struct ProfRec { hipEvent_t a, b; int kind; double work; };
std::vector<ProfRec> g_prof;
size_t g_prof_used = 0;
bool g_prof_on = false;
unsigned g_prof_mask = ~0u;
int g_prof_stride = 1;
int64_t g_prof_seen[8] = {0, 0, 0, 0, 0, 0, 0, 0};
}  // namespace

KzvProfScope::KzvProfScope(int kind, double work, hipStream_t stream) : slot(-1), s(stream) {
    if (!g_prof_on || !((g_prof_mask >> kind) & 1u)) return;
    const int64_t seen = g_prof_seen[kind & 7]++;
    if (seen % g_prof_stride || g_prof_used >= g_prof.size()) return;
    slot = (int)g_prof_used++;
    g_prof[slot].kind = kind; g_prof[slot].work = work;
    (void)hipEventRecord(g_prof[slot].a, s);
}
KzvProfScope::~KzvProfScope() {
    if (slot >= 0) (void)hipEventRecord(g_prof[slot].b, s);
}

extern "C" int kzv_prof_enable(int on, int capacity) {
    if (on) {
        while ((int)g_prof.size() < capacity) {
            ProfRec r{};
            if (hipEventCreateWithFlags(&r.a, hipEventDisableSystemFence) != hipSuccess || hipEventCreateWithFlags(&r.b, hipEventDisableSystemFence) != hipSuccess) return kzv_fail(KZV_E_HIP, "prof: event create");
            g_prof.push_back(r);
        }
        g_prof_used = 0;
        for (int i = 0; i < 8; ++i) g_prof_seen[i] = 0;
    }
    g_prof_on = on != 0;
    return KZV_OK;
}

extern "C" int kzv_prof_select(unsigned kind_mask) { g_prof_mask = kind_mask; return KZV_OK; }
extern "C" int kzv_prof_sample(int stride) {
    if (stride < 1) return kzv_fail(KZV_E_ARG, "prof_sample: stride >= 1");
    g_prof_stride = stride;
    return KZV_OK;
}
extern "C" int64_t kzv_prof_seen(int kind) { return kind >= 0 && kind < 8 ? g_prof_seen[kind] : 0; }

// Sums the recorded launches of `kind` (call after the stream is synchronised): total ms, total work units
// (FLOPs), launch count.  Does not reset; kzv_prof_enable(1, n) starts a new recording.
extern "C" int kzv_prof_collect(int kind, double* total_ms, double* total_work, int64_t* launches) {
    double ms = 0, work = 0; int64_t n = 0;
    for (size_t i = 0; i < g_prof_used; ++i) {
        if (g_prof[i].kind != kind) continue;
        float t = 0.f;
        if (hipEventElapsedTime(&t, g_prof[i].a, g_prof[i].b) != hipSuccess) return kzv_fail(KZV_E_HIP, "prof: elapsed (stream not synchronised?)");
        ms += t; work += g_prof[i].work; ++n;
    }
    if (total_ms) *total_ms = ms;
    if (total_work) *total_work = work;
    if (launches) *launches = n;
    return KZV_OK;
}

// Opt-in launch profiler (bench.py's roofline leg): HIP events bracket launches on the launch stream, accumulated per KERNEL
// CLASS together with the algorithmic work of the bracketed launches.  Off by default; the only process-global state in the
// library (the development build adds its knobs).  While it is on, the step loop launches eagerly (no graph replay) so that the brackets
// see every launch.
#include <stdlib.h>
#include "la_common.h"

#define LA_PROF_MAX 16384
static struct {
    int enabled, count, overflow;
    hipEvent_t ev0[LA_PROF_MAX], ev1[LA_PROF_MAX];
    unsigned char cls[LA_PROF_MAX];
    int created;
    double flops[LA_PC_NCLASS], bytes[LA_PC_NCLASS];
    long n[LA_PC_NCLASS];
    int stride;              // 1 = bracket every launch; k > 1 = a hashed 1-in-k sample (an event pair costs ~3 us on the stream)
    unsigned seq;            // all bracketable launches seen since la_prof_begin
} g_prof;

#ifdef LA_DEV
// dev knobs (development build only): kernel-variant selectors so that two variants can be timed in interleaved rounds of ONE process
// (cdna_hip_programming.md rule 24).  Every knob defaults to 0 = the shipped configuration.
static int g_knob[LA_NKNOB];
static const bool g_knob_env = []() {      // LA_DEV_KNOBS="id=value,id=value": initial knob values (development runs of whole test files)
    const char* e = getenv("LA_DEV_KNOBS");
    while (e && *e) {
        char* end = nullptr;
        const long id = strtol(e, &end, 10);
        if (end == e || *end != '=') break;
        const long v = strtol(end + 1, &end, 10);
        if (id >= 0 && id < LA_NKNOB) g_knob[id] = (int)v;
        e = *end == ',' ? end + 1 : end;
        if (*end != ',') break;
    }
    return true;
}();
int la_dev_knob(int id) { return id >= 0 && id < LA_NKNOB ? g_knob[id] : 0; }
const char* la_dev_env(const char* name) { return getenv(name); }
extern "C" int la_dev_knob_set(int id, int value) {
    LA_CHECK_ARG(id >= 0 && id < LA_NKNOB, "dev_knob_set: unknown knob");
    g_knob[id] = value;
    return LA_OK;
}
#endif

extern "C" int la_prof_set_stride(int stride) {
    LA_CHECK_ARG(stride >= 1 && stride <= 64, "prof: stride must be 1..64");
    g_prof.stride = stride;
    return LA_OK;
}
extern "C" long la_prof_total_launches(void) { return (long)g_prof.seq; }
bool la_prof_enabled() { return g_prof.enabled != 0; }

extern "C" int la_prof_begin(void) {
    if (g_prof.created < LA_PROF_MAX) {
        for (int i = g_prof.created; i < LA_PROF_MAX; ++i) {
            LA_HIP(hipEventCreate(&g_prof.ev0[i]));
            LA_HIP(hipEventCreate(&g_prof.ev1[i]));
            g_prof.created = i + 1;
        }
    }
    g_prof.count = 0; g_prof.overflow = 0; g_prof.seq = 0;
    for (int c = 0; c < LA_PC_NCLASS; ++c) { g_prof.flops[c] = g_prof.bytes[c] = 0; g_prof.n[c] = 0; }
    if (g_prof.stride < 1) g_prof.stride = 1;
    g_prof.enabled = 1;
    return LA_OK;
}

// bracket start: returns a slot (>= 0) or -1 when this launch is not sampled / the profiler is off
int la_prof_open(int cls, double flops, double bytes, hipStream_t stream) {
    if (!g_prof.enabled || cls < 0 || cls >= LA_PC_NCLASS) return -1;
    // (sampling is by a hash of the launch sequence number, so that no periodic launch pattern can alias with it)
    const unsigned s = g_prof.seq++;
    if (g_prof.stride > 1 && ((s * 2654435761u) >> 13) % (unsigned)g_prof.stride != 0) return -1;
    if (g_prof.count >= LA_PROF_MAX) { g_prof.overflow = 1; return -1; }
    const int slot = g_prof.count++;
    g_prof.cls[slot] = (unsigned char)cls;
    g_prof.flops[cls] += flops; g_prof.bytes[cls] += bytes; g_prof.n[cls] += 1;
    if (hipEventRecord(g_prof.ev0[slot], stream) != hipSuccess) { --g_prof.count; return -1; }
    return slot;
}
void la_prof_close(int slot, hipStream_t stream) {
    if (slot >= 0) (void)hipEventRecord(g_prof.ev1[slot], stream);
}

static int collect(double* ms_by_class) {
    g_prof.enabled = 0;
    for (int c = 0; c < LA_PC_NCLASS; ++c) ms_by_class[c] = 0;
    for (int i = 0; i < g_prof.count; ++i) {
        LA_HIP(hipEventSynchronize(g_prof.ev1[i]));
        float t = 0.f;
        LA_HIP(hipEventElapsedTime(&t, g_prof.ev0[i], g_prof.ev1[i]));
        ms_by_class[g_prof.cls[i]] += t;
    }
    return LA_OK;
}

// per-class totals since la_prof_begin: arrays of LA_PC_NCLASS entries (see la_common.h for the class ids)
extern "C" int la_prof_end_classes(double* ms, long* launches, double* flops, double* bytes, int nclass) {
    LA_CHECK_ARG(ms && launches && flops && bytes && nclass == LA_PC_NCLASS, "prof_end_classes: arrays of LA_PC_NCLASS entries expected");
    int rc = collect(ms);
    if (rc) return rc;
    for (int c = 0; c < LA_PC_NCLASS; ++c) { launches[c] = g_prof.n[c]; flops[c] = g_prof.flops[c]; bytes[c] = g_prof.bytes[c]; }
    return g_prof.overflow ? LA_ERR_WORKSPACE : LA_OK;
}
extern "C" int la_prof_num_classes(void) { return LA_PC_NCLASS; }

// total device time (ms), launches, algorithmic FLOPs and algorithmic bytes of the CONTRACTION launches since la_prof_begin
extern "C" int la_prof_end(double* total_ms, long* launches, double* flops, double* bytes) {
    double ms[LA_PC_NCLASS];
    int rc = collect(ms);
    if (rc) return rc;
    double t = 0, f = 0, b = 0; long n = 0;
    for (int c = LA_PC_CONV_HALO; c <= LA_PC_CONV_F32; ++c) { t += ms[c]; f += g_prof.flops[c]; b += g_prof.bytes[c]; n += g_prof.n[c]; }
    if (total_ms) *total_ms = t;
    if (launches) *launches = n;
    if (flops) *flops = f;
    if (bytes) *bytes = b;
    return g_prof.overflow ? LA_ERR_WORKSPACE : LA_OK;
}

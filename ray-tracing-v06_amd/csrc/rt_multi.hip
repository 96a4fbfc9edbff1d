// rt_multi.hip — rt_multi_renderer_*: one host process drives the N GPUs of a node; one grouped RCCL exchange at frame end.
#include "rt_runtime.hpp"

// ---------------------------------------------------------------------------------------------
// Multi-GPU renderer: ONE host process drives the N GPUs of a node (SURVEY.md §5 last row, §8e).  Rank i = device i renders
// the tiles t with t % N == i into its compact shard; at frame end ONE grouped RCCL exchange moves the shards to device 0
// over xGMI (every peer has its own link to the root, so the N - 1 transfers run side by side), and assemble_kernel
// de-interleaves them into the row-major image there.  Nothing else is communicated: the scene is replicated (tens of KB)
// and the RNG is keyed by global pixel and sample, so the image has the same bits for every N.
// RCCL is bound at first use (dlopen of librccl.so.1: ncclCommInitAll, ncclGroupStart/End, ncclSend, ncclRecv,
// ncclCommDestroy, ncclGetErrorString) so that single-GPU callers do not map the 570-MB collective library.
// ---------------------------------------------------------------------------------------------
namespace {
struct RcclApi {
    void* handle = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
};
RcclApi g_rccl;
std::once_flag g_rccl_once;
int g_rccl_rc = RT_OK;
std::string g_rccl_error;

int rccl_bind_once() {
    void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) return rt_fail(RT_ERR_HIP, "multi-GPU rendering needs RCCL: %s", dlerror());
    RcclApi a;
    a.handle = h;
#define RT_RCCL_SYM(field, name)                                                                    \
    a.field = reinterpret_cast<decltype(a.field)>(dlsym(h, name));                                  \
    if (!a.field) return rt_fail(RT_ERR_HIP, "librccl.so.1 has no symbol %s", name)
    RT_RCCL_SYM(CommInitAll, "ncclCommInitAll");
    RT_RCCL_SYM(CommDestroy, "ncclCommDestroy");
    RT_RCCL_SYM(GroupStart, "ncclGroupStart");
    RT_RCCL_SYM(GroupEnd, "ncclGroupEnd");
    RT_RCCL_SYM(Send, "ncclSend");
    RT_RCCL_SYM(Recv, "ncclRecv");
    RT_RCCL_SYM(GetErrorString, "ncclGetErrorString");
#undef RT_RCCL_SYM
    g_rccl = a;
    return RT_OK;
}
// bound once per process, whichever thread creates the first multi-GPU renderer
int rccl_bind() {
    std::call_once(g_rccl_once, [] {
        g_rccl_rc = rccl_bind_once();
        if (g_rccl_rc != RT_OK) g_rccl_error = rt_last_error();
    });
    return g_rccl_rc == RT_OK ? RT_OK : rt_fail(g_rccl_rc, "%s", g_rccl_error.c_str());
}
}  // namespace

#define RCCL_TRY(expr)                                                                                                  \
    do {                                                                                                                \
        ncclResult_t _r = (expr);                                                                                       \
        if (_r != ncclSuccess) return rt_fail(RT_ERR_HIP, "%s failed: %s (%s:%d)", #expr, g_rccl.GetErrorString(_r), __FILE__, __LINE__); \
    } while (0)

// How the shards travel to devices[0] at frame end.
//   RT_TRANSPORT_RCCL   (default): one grouped ncclSend / ncclRecv exchange over xGMI, one rank per GPU.
//   RT_TRANSPORT_MEMCPY (RT06_MULTI_TRANSPORT=memcpy; tests and single-GPU boxes): hipMemcpyAsync on the ranks' own streams, ordered
//       by events.  It lifts the one-rank-per-device rule, so N ranks can share ONE GPU and the whole N > 1 branch — shard offsets,
//       stream ordering, assemble_kernel, download — runs where RCCL would refuse (it does not accept two ranks on one device).
enum : uint32_t { RT_TRANSPORT_RCCL = 0, RT_TRANSPORT_MEMCPY = 1 };

struct rt_multi_renderer {
    uint32_t width = 0, height = 0;
    uint32_t transport = RT_TRANSPORT_RCCL;
    std::vector<int> devices;
    std::vector<rt_renderer*> parts;     // parts[i]: rank i of N on devices[i]
    std::vector<ncclComm_t> comms;
    std::vector<hipEvent_t> ev_part;     // per rank, on its device: its render is enqueued / (memcpy transport) its shard has been copied
    DevBuf gathered, image;              // on devices[0]: N shards back to back; the assembled row-major frame
    hipEvent_t ev_rendered = nullptr, ev_done = nullptr;   // on devices[0]'s stream: every rank has rendered / after the assembly
    size_t shard_floats = 0;             // floats of one rank's tile-major shard (the same on every rank)
    float last_total_ms = 0.0f;
    bool rendered = false;
    ~rt_multi_renderer() {
        for (ncclComm_t c : comms) if (c) (void)g_rccl.CommDestroy(c);
        for (size_t i = 0; i < ev_part.size(); i++) {
            if (!ev_part[i]) continue;
            (void)hipSetDevice(devices[i]);
            (void)hipEventDestroy(ev_part[i]);
        }
        for (rt_renderer* r : parts) rt_renderer_destroy(r);
        if (!devices.empty()) (void)hipSetDevice(devices[0]);
        if (ev_rendered) (void)hipEventDestroy(ev_rendered);
        if (ev_done) (void)hipEventDestroy(ev_done);
    }
};

extern "C" int rt_multi_renderer_create(const rt_render_config* cfg, const rt_camera* cam, const rt_world_flat* world, uint32_t n_gpus,
                                        const int32_t* devices, rt_multi_renderer** out) {
    if (!cfg || !cam || !world || !out) return rt_fail(RT_ERR_INVALID, "rt_multi_renderer_create: null argument");
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev == 0)
        return rt_fail(RT_ERR_NO_DEVICE, "no HIP device available: the HIP path is required, there is no CPU fallback");
    uint32_t transport = RT_TRANSPORT_RCCL;
    if (const char* env = std::getenv("RT06_MULTI_TRANSPORT")) {
        if (std::strcmp(env, "memcpy") == 0) transport = RT_TRANSPORT_MEMCPY;
        else if (std::strcmp(env, "rccl") != 0) return rt_fail(RT_ERR_INVALID, "RT06_MULTI_TRANSPORT=%s: expected rccl or memcpy", env);
    }
    if (n_gpus == 0 || n_gpus > 64) return rt_fail(RT_ERR_INVALID, "rt_multi_renderer_create: %u ranks asked for", n_gpus);
    if (transport == RT_TRANSPORT_RCCL && (int)n_gpus > n_dev) return rt_fail(RT_ERR_INVALID, "rt_multi_renderer_create: %u GPUs asked for, %d present", n_gpus, n_dev);
    std::vector<int> devs(n_gpus);
    for (uint32_t i = 0; i < n_gpus; i++) {
        devs[i] = devices ? devices[i] : (transport == RT_TRANSPORT_MEMCPY ? (int)(i % (uint32_t)n_dev) : (int)i);
        if (devs[i] < 0 || devs[i] >= n_dev) return rt_fail(RT_ERR_INVALID, "rt_multi_renderer_create: device %d out of range", devs[i]);
        for (uint32_t j = 0; j < i && transport == RT_TRANSPORT_RCCL; j++)
            if (devs[j] == devs[i]) return rt_fail(RT_ERR_INVALID, "rt_multi_renderer_create: device %d listed twice (one rank per GPU)", devs[i]);
    }
    int rc = transport == RT_TRANSPORT_RCCL ? rccl_bind() : RT_OK;
    if (rc != RT_OK) return rc;
    std::unique_ptr<rt_multi_renderer> m(new rt_multi_renderer());
    m->width = cfg->width; m->height = cfg->height;
    m->transport = transport;
    m->devices = devs;
    for (uint32_t i = 0; i < n_gpus; i++) {
        rt_render_config c = *cfg;
        c.device = devs[i]; c.rank = i; c.world_size = n_gpus;
        rt_renderer* r = nullptr;
        rc = rt_renderer_create(&c, cam, world, &r);
        if (rc != RT_OK) return rc;
        m->parts.push_back(r);
    }
    if (transport == RT_TRANSPORT_RCCL) {
        m->comms.assign(n_gpus, nullptr);
        RCCL_TRY(g_rccl.CommInitAll(m->comms.data(), (int)n_gpus, devs.data()));
    }
    m->ev_part.assign(n_gpus, nullptr);
    for (uint32_t i = 0; i < n_gpus; i++) {
        HIP_TRY(hipSetDevice(devs[i]));
        HIP_TRY(hipEventCreateWithFlags(&m->ev_part[i], hipEventDisableTiming));
    }
    HIP_TRY(hipSetDevice(devs[0]));
    const size_t image_floats = (size_t)cfg->width * cfg->height * 4;
    HIP_TRY(m->image.alloc(image_floats * sizeof(float)));
    rc = rt_renderer_shard_floats(m->parts[0], &m->shard_floats);
    if (rc != RT_OK) return rc;
    if (n_gpus > 1) HIP_TRY(m->gathered.alloc(m->shard_floats * n_gpus * sizeof(float)));
    HIP_TRY(hipEventCreate(&m->ev_rendered));
    HIP_TRY(hipEventCreate(&m->ev_done));
    *out = m.release();
    return RT_OK;
}

extern "C" void rt_multi_renderer_destroy(rt_multi_renderer* m) { delete m; }

// a failure between the launches and the final synchronisation must not leave work in flight on the ranks' streams
static int multi_fail_drain(rt_multi_renderer* m, int rc) {
    const std::string msg = rt_last_error();   // the drains below may overwrite the message of the failure we report
    for (size_t i = 0; i < m->parts.size(); i++)
        if (hipSetDevice(m->devices[i]) == hipSuccess) (void)hipStreamSynchronize(rt_renderer_own_stream(m->parts[i]));
    return rt_fail(rc, "%s", msg.c_str());
}

// launches, exchange, assembly, final synchronisation; ANY early return leaves work enqueued on some rank's stream and goes through
// multi_fail_drain in the caller below
static int multi_render_body(rt_multi_renderer* m) {
    const uint32_t n = (uint32_t)m->parts.size();
    const auto t0 = std::chrono::steady_clock::now();
    hipStream_t s0 = rt_renderer_own_stream(m->parts[0]);
    for (uint32_t i = 0; i < n; i++) {   // every GPU renders its tiles; the launches are asynchronous, so the N kernels run side by side
        int rc = rt_renderer_render_async(m->parts[i], rt_renderer_own_stream(m->parts[i]), nullptr);
        if (rc != RT_OK) return rc;
        HIP_TRY(hipEventRecord(m->ev_part[i], rt_renderer_own_stream(m->parts[i])));
    }
    // stream 0 waits for EVERY rank's render before the exchange timer starts: times()[2] is then exchange + assembly, not the
    // slowest rank's tail (the receive would otherwise absorb the imbalance of the frame)
    HIP_TRY(hipSetDevice(m->devices[0]));
    for (uint32_t i = 1; i < n; i++) HIP_TRY(hipStreamWaitEvent(s0, m->ev_part[i], 0));
    HIP_TRY(hipEventRecord(m->ev_rendered, s0));
    // the single frame-end exchange: rank i sends its shard to rank 0 (rank 0 to itself), rank 0 receives N shards in rank order.
    // With one GPU this is the degenerate self-exchange of the whole row-major frame.
    const size_t count = n > 1 ? m->shard_floats : (size_t)m->width * m->height * 4;
    float* dst = n > 1 ? m->gathered.as<float>() : m->image.as<float>();
    if (m->transport == RT_TRANSPORT_RCCL) {
        // an error inside the group still CLOSES the group (an open group makes the process's next collective call hang)
        ncclResult_t first = g_rccl.GroupStart();
        const char* what = "ncclGroupStart";
        if (first == ncclSuccess) {
            for (uint32_t i = 0; i < n && first == ncclSuccess; i++) {
                first = g_rccl.Send(rt_renderer_own_framebuffer(m->parts[i]), count, ncclFloat, 0, m->comms[i], rt_renderer_own_stream(m->parts[i]));
                what = "ncclSend";
            }
            for (uint32_t i = 0; i < n && first == ncclSuccess; i++) {
                first = g_rccl.Recv(dst + (size_t)i * count, count, ncclFloat, (int)i, m->comms[0], s0);
                what = "ncclRecv";
            }
            const ncclResult_t end = g_rccl.GroupEnd();
            if (first == ncclSuccess && end != ncclSuccess) { first = end; what = "ncclGroupEnd"; }
        }
        if (first != ncclSuccess) {
            return rt_fail(RT_ERR_HIP, "%s failed in the frame-end exchange: %s", what, g_rccl.GetErrorString(first));
        }
    } else {
        for (uint32_t i = 0; i < n; i++) {
            HIP_TRY(hipSetDevice(m->devices[i]));
            HIP_TRY(hipMemcpyAsync(dst + (size_t)i * count, rt_renderer_own_framebuffer(m->parts[i]), count * sizeof(float), hipMemcpyDeviceToDevice, rt_renderer_own_stream(m->parts[i])));
            HIP_TRY(hipEventRecord(m->ev_part[i], rt_renderer_own_stream(m->parts[i])));
        }
        HIP_TRY(hipSetDevice(m->devices[0]));
        for (uint32_t i = 1; i < n; i++) HIP_TRY(hipStreamWaitEvent(s0, m->ev_part[i], 0));
    }
    HIP_TRY(hipSetDevice(m->devices[0]));
    if (n > 1) {
        int rc = rt_renderer_assemble(m->parts[0], m->gathered.as<float>(), m->image.as<float>(), s0);
        if (rc != RT_OK) return rc;
    }
    HIP_TRY(hipEventRecord(m->ev_done, s0));
    for (uint32_t i = 0; i < n; i++) {
        HIP_TRY(hipSetDevice(m->devices[i]));
        HIP_TRY(hipStreamSynchronize(rt_renderer_own_stream(m->parts[i])));
    }
    // RT_TRAVERSAL_QUEUE / _WIDE4 worlds: an overflow of the 32 entries is an error here too (and a ray-exchange protocol error).  EVERY rank's
    // flags are read (and thereby cleared); the first failure is the one reported
    int first_rc = RT_OK;
    std::string first_msg;
    for (uint32_t i = 0; i < n; i++) {
        HIP_TRY(hipSetDevice(m->devices[i]));
        const int rc = rt_renderer_check_device_flags(m->parts[i]);
        if (rc != RT_OK && first_rc == RT_OK) { first_rc = rc; first_msg = rt_last_error(); }
    }
    if (first_rc != RT_OK) return rt_fail(first_rc, "%s", first_msg.c_str());
    m->last_total_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
    m->rendered = true;
    return RT_OK;
}

extern "C" int rt_multi_renderer_render(rt_multi_renderer* m) {
    if (!m) return rt_fail(RT_ERR_INVALID, "rt_multi_renderer_render: null renderer");
    const int rc = multi_render_body(m);
    return rc == RT_OK ? RT_OK : multi_fail_drain(m, rc);   // nothing stays in flight behind an error, whichever call failed
}

extern "C" int rt_multi_renderer_download(rt_multi_renderer* m, float* host_rgba, size_t n_floats) {
    if (!m || !host_rgba) return rt_fail(RT_ERR_INVALID, "rt_multi_renderer_download: null argument");
    const size_t need = (size_t)m->width * m->height * 4;
    if (n_floats != need) return rt_fail(RT_ERR_INVALID, "rt_multi_renderer_download: buffer holds %zu floats, image needs %zu", n_floats, need);
    HIP_TRY(hipSetDevice(m->devices[0]));
    HIP_TRY(hipStreamSynchronize(rt_renderer_own_stream(m->parts[0])));
    HIP_TRY(hipMemcpy(host_rgba, m->image.p, need * sizeof(float), hipMemcpyDeviceToHost));
    return RT_OK;
}

extern "C" int rt_multi_renderer_times(rt_multi_renderer* m, float out_ms[3]) {
    if (!m || !out_ms) return rt_fail(RT_ERR_INVALID, "rt_multi_renderer_times: null argument");
    if (!m->rendered) return rt_fail(RT_ERR_INVALID, "rt_multi_renderer_times: nothing rendered yet");
    out_ms[0] = m->last_total_ms;
    float worst = 0.0f;
    for (rt_renderer* r : m->parts) {
        float ms = 0.0f;
        int rc = rt_renderer_last_kernel_ms(r, &ms);
        if (rc != RT_OK) return rc;
        worst = std::max(worst, ms);
    }
    out_ms[1] = worst;
    HIP_TRY(hipSetDevice(m->devices[0]));
    HIP_TRY(hipEventElapsedTime(out_ms + 2, m->ev_rendered, m->ev_done));
    return RT_OK;
}

extern "C" int rt_multi_renderer_gpus(const rt_multi_renderer* m, uint32_t* out) {
    if (!m || !out) return rt_fail(RT_ERR_INVALID, "rt_multi_renderer_gpus: null argument");
    *out = (uint32_t)m->parts.size();
    return RT_OK;
}


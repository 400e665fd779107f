// qv_capi.hip -- C ABI entry points (include/qverse.h), engine lifetime, table upload.

#include "qv_common.h"
#include "qv_kernels.h"
#include "qv_layers.h"

#include <math.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <chrono>
#include <fstream>
#include <mutex>

static std::string g_create_error;

// ---- how many streams does the runtime actually run side by side? ---------------------------------------------------
// The HIP runtime maps streams onto GPU_MAX_HW_QUEUES hardware queues (default 4), read ONCE when HIP initialises;
// streams that share a queue serialise.  An engine with four batches in flight needs a queue per context stream plus the
// caller's (14.8 k utt/s on 4 queues against 17.3 k on 8, profiles/r02_h_contexts_hwq_sweep.txt), and a host that touched
// HIP before the variable was set silently gets the default.  Instead of trusting the environment the library measures:
// four fresh streams -- what an engine with four contexts is about to create; the caller's stream and RCCL's already
// exist and hold their queues -- each run a one-wave kernel that spins for a fixed
// wall-clock time; the elapsed time over the spin time is how many of them shared a queue.  Measured
// (tools/hwq_probe.py, profiles/archive/r03_c_hwq_probe.txt): with GPU_MAX_HW_QUEUES=8 in place 7 fresh streams run side by side in
// a plain process and 4 after RCCL has created its own; on the default 4 queues only 3 do.
namespace {
__global__ void k_spin(long long ticks) {   // wall_clock64: 100 MHz
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(16);
}
}  // namespace

#define QV_PROBE_STREAMS 4
// rounds = elapsed / spin time of `ns` spin kernels on `ns` fresh streams: 1 = all side by side
static double probe_rounds(int ns) {
    constexpr long long SPIN_TICKS = 40000;   // 400 us
    std::vector<hipStream_t> st(ns, nullptr);
    for (int i = 0; i < ns; ++i)
        if (hipStreamCreateWithFlags(&st[i], hipStreamNonBlocking) != hipSuccess) return 0.0;
    for (int i = 0; i < ns; ++i) hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, st[i], 1LL);   // warm-up: queue creation, code load
    for (int i = 0; i < ns; ++i) (void)hipStreamSynchronize(st[i]);
    double best = 1e30;
    for (int rep = 0; rep < 2; ++rep) {
        const auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < ns; ++i) hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, st[i], SPIN_TICKS);
        for (int i = 0; i < ns; ++i) (void)hipStreamSynchronize(st[i]);
        best = std::min(best, std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count());
    }
    for (int i = 0; i < ns; ++i) (void)hipStreamDestroy(st[i]);
    return best / (SPIN_TICKS / 100.0);
}

// host calls on one engine are serialised (qv_engine::mu); a null engine falls through to the entry point's own check
#define QV_SERIALISE(e) std::unique_lock<std::recursive_mutex> qv_lock_; if (e) qv_lock_ = std::unique_lock<std::recursive_mutex>((e)->mu)

// Device-side order between consecutive calls on different caller streams (two host threads, each with its own stream,
// share the engine's workspace): wait for the previous call's tail on entry, leave a new tail on exit.
namespace {
struct QvStreamOrder {
    qv_engine *e;
    hipStream_t s;
    QvStreamOrder(qv_engine *e_, hipStream_t s_) : e(e_), s(s_) {
        if (e && e->tail_valid && e->tail_stream != s) (void)hipStreamWaitEvent(s, e->tail_ev, 0);
    }
    ~QvStreamOrder() {
        if (!e) return;
        if (!e->tail_ev && hipEventCreateWithFlags(&e->tail_ev, hipEventDisableTiming) != hipSuccess) { e->tail_ev = nullptr; return; }
        if (hipEventRecord(e->tail_ev, s) == hipSuccess) { e->tail_stream = s; e->tail_valid = true; }
    }
};
}  // namespace
#define QV_ORDERED(e, stream) QvStreamOrder qv_order_((e), (hipStream_t)(stream))

extern "C" int32_t qv_probe_concurrent_streams(void) {
    // one probe per process, also when two threads create engines at once: the mutex covers the measurement itself (two
    // probes side by side would halve each other's figure), and a failed probe (0) is not cached
    static std::mutex mu;
    static int cached = 0;
    std::lock_guard<std::mutex> lk(mu);
    if (cached) return cached;
    const double rounds = probe_rounds(QV_PROBE_STREAMS);
    if (rounds == 0.0) return 0;
    cached = rounds < 1.5 ? QV_PROBE_STREAMS : rounds < 2.5 ? QV_PROBE_STREAMS / 2 : 1;
    return cached;
}

// dev tool (tools/hwq_probe.py): the raw figure for any number of streams, not cached
extern "C" double qv_debug_probe_rounds(int32_t n_streams) { return n_streams >= 1 && n_streams <= 32 ? probe_rounds(n_streams) : 0.0; }

static std::mutex g_create_error_mu;
void qv_set_error(qv_engine *e, const std::string &msg) {
    if (e) e->last_error = msg;   // callers hold e->mu (every entry point takes it before it can fail)
    else { std::lock_guard<std::mutex> lk(g_create_error_mu); g_create_error = msg; }
}

// The text is copied into a buffer of the CALLING thread under the engine's lock: another thread's call may replace
// (and free) the engine's string at any time, the returned pointer stays valid until this thread asks again.
extern "C" const char *qv_last_error(const qv_engine *e) {
    static thread_local std::string mine;
    if (e) {
        std::lock_guard<std::recursive_mutex> lk(const_cast<qv_engine *>(e)->mu);
        mine = e->last_error;
    } else {
        std::lock_guard<std::mutex> lk(g_create_error_mu);
        mine = g_create_error;
    }
    return mine.c_str();
}

extern "C" int qv_debug_int4_roundtrip(const float *w, int32_t N, int32_t K, float *out) {
    if (!w || !out || N < 64 || N % 64 != 0 || K < 128 || K % 128 != 0) return QV_ERR_ARG;
    std::vector<uint8_t> q((size_t)N * K / 2, 0);
    std::vector<half_t> sc((size_t)N * (K / 128) * 2);
    qv_pack_w4(w, N, K, q.data(), sc.data());
    // walk the device layout exactly as the kernel does (qv_kernels.h: GemmArgs::Wq / wscale)
    const int nk = K / 64;
    for (int n = 0; n < N; ++n)
        for (int k = 0; k < K; ++k) {
            int kt = k >> 6, c = (k & 63) >> 3, e = k & 7, p = e >> 1;
            const uint8_t *tile = q.data() + ((size_t)(n >> 6) * nk + kt) * 2048 + (size_t)(n & 63) * 32;
            uint32_t chunk;
            memcpy(&chunk, tile + 4 * (c ^ ((n >> 2) & 7)), 4);
            uint32_t pair = (chunk >> (4 * p)) & 0x000F000Fu;
            int code = (e & 1) ? (int)(pair >> 16) : (int)(pair & 0xFFFF);
            const half_t *sz = sc.data() + ((size_t)(k >> 7) * N + n) * 2;
            out[(size_t)n * K + k] = ((float)code - ((float)sz[1] - 1024.f)) * (float)sz[0];
        }
    return QV_OK;
}

extern "C" int qv_debug_int8_roundtrip(const float *w, int32_t N, int32_t K, float *out) {
    if (!w || !out || N < 64 || N % 64 != 0 || K < 64 || K % 64 != 0) return QV_ERR_ARG;
    std::vector<uint8_t> q((size_t)N * K, 0);
    std::vector<float> sc((size_t)N);
    qv_pack_w8(w, N, K, q.data(), sc.data());
    // walk the device layout exactly as the kernel does (qv_kernels.h: GemmArgs::W8 / w8scale)
    const int nk = K / 64;
    for (int n = 0; n < N; ++n)
        for (int k = 0; k < K; ++k) {
            int kt = k >> 6, c = (k & 63) >> 3;
            const uint8_t *row = q.data() + ((size_t)(n >> 6) * nk + kt) * 4096 + (size_t)(n & 63) * 64;
            int code = (int)row[((c ^ ((n >> 2) & 7)) << 3) + (k & 7)] - 128;
            out[(size_t)n * K + k] = (float)code * sc[n];
        }
    return QV_OK;
}

extern "C" const char *qv_build_info(void) {
    // the compiler is part of the contract: the GEMM loaders are inline-asm buffer loads with hand-counted s_waitcnt
    // vmcnt(N), correct for the register allocation THIS hipcc produced (tests/test_loader_hazards.py checks the code
    // objects of every build; a different compiler has to pass that test again)
    static char buf[256];
    snprintf(buf, sizeof buf, "libqverse gfx950 hip-%d.%d.%d clang-%s", HIP_VERSION_MAJOR, HIP_VERSION_MINOR, HIP_VERSION_PATCH,
             __clang_version__);
    return buf;
}

extern "C" void qv_config_default(qv_config *c) {
    memset(c, 0, sizeof *c);
    c->struct_size = (int32_t)sizeof(qv_config);
    c->device = 0;
    c->with_model = 1;
    c->precision = QV_PREC_FP16;
    c->max_batch = 64;
    c->max_samples = 480000;
    c->random_weights_seed = 20260630ull;
    c->top_text = 100;
    c->top_span_refs = 80;
    c->max_span = 6;
    c->threshold = 0.80;
    c->text_weight = 0.0;
    c->span_penalty = 0.5;
    c->skip_unused_passes = 1;
    c->n_contexts = 1;
}

extern "C" int32_t qv_frames_for_samples(int64_t n) {
    int64_t t = n / 160 + 1;  // STFT center=True, hop 160
    for (int i = 0; i < 3; ++i) t = (t + 2 - 3) / 2 + 1;  // conv k3 s2 p1
    return (int32_t)t;
}

// ------------------------------------------------------------------ blob reader --------
struct Blob {
    std::vector<uint8_t> data;
    const void *get(const char *name, size_t *nbytes = nullptr) const {
        uint32_t n;
        memcpy(&n, data.data() + 8, 4);
        for (uint32_t i = 0; i < n; ++i) {
            const uint8_t *e = data.data() + 16 + 40 * (size_t)i;
            if (strncmp((const char *)e, name, 24) == 0) {
                uint64_t off, nb;
                memcpy(&off, e + 24, 8);
                memcpy(&nb, e + 32, 8);
                if (off + nb > data.size()) return nullptr;
                if (nbytes) *nbytes = (size_t)nb;
                return data.data() + off;
            }
        }
        return nullptr;
    }
};

template <typename T>
static int upload(qv_engine *eng, const T *host, size_t count, const T **dev) {
    void *p = nullptr;
    QV_HIP(hipMalloc(&p, std::max<size_t>(count * sizeof(T), 16)));
    eng->allocs.push_back(p);
    if (count) QV_HIP(hipMemcpy(p, host, count * sizeof(T), hipMemcpyHostToDevice));
    *dev = (const T *)p;
    return QV_OK;
}

template <typename T>
static int dalloc(qv_engine *eng, size_t count, T **dev) {
    void *p = nullptr;
    QV_HIP(hipMalloc(&p, std::max<size_t>(count * sizeof(T), 16)));
    QV_HIP(hipMemset(p, 0, std::max<size_t>(count * sizeof(T), 16)));
    eng->allocs.push_back(p);
    *dev = (T *)p;
    return QV_OK;
}

#define QV_TRY(x) do { int rc_ = (x); if (rc_) return rc_; } while (0)

static int load_tables(qv_engine *eng, const char *path) {
    Blob blob;
    {
        std::ifstream f(path, std::ios::binary | std::ios::ate);
        if (!f) { qv_set_error(eng, std::string("cannot open tables file: ") + (path ? path : "(null)")); return QV_ERR_IO; }
        std::streamsize sz = f.tellg();
        f.seekg(0);
        blob.data.resize((size_t)sz);
        if (!f.read((char *)blob.data.data(), sz) || sz < 16 || memcmp(blob.data.data(), "QVTB0001", 8)) {
            qv_set_error(eng, "tables file malformed (bad magic)");
            return QV_ERR_IO;
        }
    }
    const char *need[] = {"meta", "surah", "ayah", "surah_start", "surah_len", "clean_off", "clean", "alt_off", "alt",
                          "nobsm_off", "nobsm", "clean_nw", "alt_nw", "nobsm_nw", "tok_off", "tok", "piece_off",
                          "piece_codes", "tri_keys", "tri_idf", "vtri_off", "vtri"};
    for (const char *n : need)
        if (!blob.get(n)) { qv_set_error(eng, std::string("tables file lacks section ") + n); return QV_ERR_IO; }
    const int32_t *meta = (const int32_t *)blob.get("meta");
    int N = meta[0], NS = meta[1], K = meta[2], NT = meta[6];
    if (K > QV_NSYM || meta[3] != QV_MAX_SPAN || meta[4] != QV_VOCAB) {
        qv_set_error(eng, "tables file incompatible (alphabet/max_span/vocab)");
        return QV_ERR_IO;
    }
    QvTables &t = eng->tab;
    t.n_verses = N; t.n_surah = NS; t.n_tri = NT;
    const uint8_t *surah = (const uint8_t *)blob.get("surah");
    const uint16_t *ayah = (const uint16_t *)blob.get("ayah");
    eng->h_surah.assign(surah, surah + N);
    eng->h_ayah.assign(ayah, ayah + N);
    QV_TRY(upload(eng, surah, N, &t.surah));
    QV_TRY(upload(eng, ayah, N, &t.ayah));
    QV_TRY(upload(eng, (const int32_t *)blob.get("surah_start"), NS + 1, &t.surah_start));
    QV_TRY(upload(eng, (const int32_t *)blob.get("surah_len"), NS, &t.surah_len));
    const uint32_t *coff = (const uint32_t *)blob.get("clean_off"), *aoff = (const uint32_t *)blob.get("alt_off"),
                   *noff = (const uint32_t *)blob.get("nobsm_off");
    const uint8_t *ctxt = (const uint8_t *)blob.get("clean"), *atxt = (const uint8_t *)blob.get("alt"),
                  *ntxt = (const uint8_t *)blob.get("nobsm");
    // padded clean array: verse texts separated by one space
    std::vector<uint8_t> cpad;
    std::vector<uint32_t> cpo(N), apo(N);
    std::vector<uint16_t> clen(N), alen(N), nlen(N);
    std::vector<int32_t> nrank(N, -1);
    int n_nobsm = 0;
    for (int v = 0; v < N; ++v) {
        cpo[v] = (uint32_t)cpad.size();
        clen[v] = (uint16_t)(coff[v + 1] - coff[v]);
        cpad.insert(cpad.end(), ctxt + coff[v], ctxt + coff[v + 1]);
        cpad.push_back(0);
        apo[v] = aoff[v];
        alen[v] = (uint16_t)(aoff[v + 1] - aoff[v]);
        nlen[v] = (uint16_t)(noff[v + 1] - noff[v]);
        if (nlen[v]) {
            nrank[v] = n_nobsm++;
            // the no-bismillah text must be a suffix of the clean text (it is clean[len(BSM):].strip())
            if (nlen[v] > clen[v] || memcmp(ntxt + noff[v], ctxt + coff[v + 1] - nlen[v], nlen[v]) != 0) {
                qv_set_error(eng, "tables: no_bsm text is not a suffix of text_clean");
                return QV_ERR_IO;
            }
        }
    }
    {
        std::vector<uint8_t> c8;
        std::vector<uint32_t> o8(N + 1);
        for (int v = 0; v < N; ++v) {
            o8[v] = (uint32_t)c8.size();
            c8.push_back(0);
            c8.insert(c8.end(), ctxt + coff[v], ctxt + coff[v + 1]);
            while (c8.size() & 7) c8.push_back(0xFF);
        }
        o8[N] = (uint32_t)c8.size();
        c8.resize(c8.size() + 64, 0xFF);
        QV_TRY(upload(eng, c8.data(), c8.size(), &t.clean8));
        QV_TRY(upload(eng, o8.data(), (size_t)N + 1, &t.clean8_off));
    }
    cpad.resize(cpad.size() + 64, 0);
    QV_TRY(upload(eng, cpad.data(), cpad.size(), &t.clean));
    QV_TRY(upload(eng, cpo.data(), (size_t)N, &t.clean_off));
    QV_TRY(upload(eng, clen.data(), (size_t)N, &t.clean_len));
    QV_TRY(upload(eng, nlen.data(), (size_t)N, &t.nobsm_len));
    {
        std::vector<uint8_t> apad(atxt, atxt + aoff[N]);
        apad.resize(apad.size() + 64, 0);  // texts are read 8 bytes at a time
        QV_TRY(upload(eng, apad.data(), apad.size(), &t.alt));
    }
    QV_TRY(upload(eng, apo.data(), (size_t)N, &t.alt_off));
    QV_TRY(upload(eng, alen.data(), (size_t)N, &t.alt_len));
    QV_TRY(upload(eng, (const uint16_t *)blob.get("clean_nw"), (size_t)N, &t.nw[0]));
    QV_TRY(upload(eng, (const uint16_t *)blob.get("alt_nw"), (size_t)N, &t.nw[1]));
    QV_TRY(upload(eng, (const uint16_t *)blob.get("nobsm_nw"), (size_t)N, &t.nw[2]));
    QV_TRY(upload(eng, nrank.data(), (size_t)N, &t.nobsm_rank));
    {   // word ends of the clean texts (the verse tracker's prefix scores)
        std::vector<uint32_t> wo(N + 1, 0);
        std::vector<uint16_t> we;
        const uint16_t *cnw = (const uint16_t *)blob.get("clean_nw");
        for (int v = 0; v < N; ++v) {
            wo[v] = (uint32_t)we.size();
            for (int i = 0; i < clen[v]; ++i)
                if (ctxt[coff[v] + i] == 0) we.push_back((uint16_t)i);
            we.push_back(clen[v]);
            if ((int)(we.size() - wo[v]) != cnw[v]) { qv_set_error(eng, "tables: clean_nw disagrees with the text"); return QV_ERR_IO; }
        }
        wo[N] = (uint32_t)we.size();
        QV_TRY(upload(eng, wo.data(), wo.size(), &t.wend_off));
        QV_TRY(upload(eng, we.data(), we.size(), &t.wend));
        std::vector<int32_t> order(N);
        for (int v = 0; v < N; ++v) order[v] = v;
        std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return clen[a] > clen[b]; });
        QV_TRY(upload(eng, order.data(), order.size(), &t.len_order));
    }
    // per-text match masks
    t.n_text = 2 * N + n_nobsm;
    std::vector<uint32_t> pmo(t.n_text);
    std::vector<uint64_t> pmv;
    auto add_text = [&](int tid, const uint8_t *s, int len) {
        int stride = qv_tmpl_w((len + 63) / 64);
        pmo[tid] = (uint32_t)pmv.size();
        pmv.resize(pmv.size() + (size_t)QV_NSYM * stride, 0ull);
        uint64_t *base = pmv.data() + pmo[tid];
        for (int i = 0; i < len; ++i)
            if (s[i] < QV_NSYM) base[(size_t)s[i] * stride + (i >> 6)] |= 1ull << (i & 63);
    };
    for (int v = 0; v < N; ++v) {
        add_text(v, ctxt + coff[v], clen[v]);
    }
    for (int v = 0; v < N; ++v) add_text(N + v, atxt + aoff[v], alen[v]);
    for (int v = 0; v < N; ++v)
        if (nlen[v]) add_text(2 * N + nrank[v], ntxt + noff[v], nlen[v]);
    pmv.resize(pmv.size() + 16, 0ull);
    QV_TRY(upload(eng, pmv.data(), pmv.size(), &t.pmv));
    QV_TRY(upload(eng, pmo.data(), pmo.size(), &t.pmv_off));
    QV_TRY(upload(eng, (const uint32_t *)blob.get("tri_keys"), (size_t)NT, &t.tri_keys));
    QV_TRY(upload(eng, (const double *)blob.get("tri_idf"), (size_t)NT, &t.tri_idf));
    const uint32_t *vto = (const uint32_t *)blob.get("vtri_off");
    QV_TRY(upload(eng, vto, (size_t)N + 1, &t.vtri_off));
    const uint16_t *vt = (const uint16_t *)blob.get("vtri");
    QV_TRY(upload(eng, vt, (size_t)vto[N], &t.vtri));
    {   // inverted trigram index, split in four verse-range slices (one per wave of k_trigram)
        std::vector<uint32_t> start(NT + 1, 0);
        for (uint32_t p = 0; p < vto[N]; ++p) {
            if (vt[p] >= NT) { qv_set_error(eng, "tables: trigram id out of range"); return QV_ERR_IO; }
            ++start[vt[p] + 1];
        }
        for (int i = 0; i < NT; ++i) start[i + 1] += start[i];
        std::vector<uint16_t> post(vto[N] + 64, 0);
        std::vector<uint32_t> fill(start.begin(), start.end() - 1), slice((size_t)NT * 5);
        for (int i = 0; i < NT; ++i) slice[(size_t)i * 5] = start[i];
        int w = 0;
        for (int v = 0; v < N; ++v) {
            while (w < 3 && v >= (int)((long long)(w + 1) * N / 4)) {   // verse v opens slice w + 1
                ++w;
                for (int i = 0; i < NT; ++i) slice[(size_t)i * 5 + w] = fill[i];
            }
            for (uint32_t p = vto[v]; p < vto[v + 1]; ++p) post[fill[vt[p]]++] = (uint16_t)v;
        }
        for (++w; w <= 4; ++w)
            for (int i = 0; i < NT; ++i) slice[(size_t)i * 5 + w] = fill[i];
        QV_TRY(upload(eng, post.data(), post.size(), &t.tri_post));
        std::vector<uint16_t> map((size_t)1 << 18, 0xFFFF);
        const uint32_t *keys = (const uint32_t *)blob.get("tri_keys");
        for (int i = 0; i < NT; ++i) {
            if (keys[i] >= map.size() || NT >= 0xFFFF) { qv_set_error(eng, "tables: trigram key out of range"); return QV_ERR_IO; }
            map[keys[i]] = (uint16_t)i;
        }
        QV_TRY(upload(eng, map.data(), map.size(), &t.tri_map));
        QV_TRY(upload(eng, slice.data(), slice.size(), &t.tri_slice));
    }
    const uint32_t *to = (const uint32_t *)blob.get("tok_off");
    QV_TRY(upload(eng, to, (size_t)N * QV_MAX_SPAN + 1, &t.tok_off));
    QV_TRY(upload(eng, (const uint16_t *)blob.get("tok"), (size_t)to[(size_t)N * QV_MAX_SPAN], &t.tok));
    {   // prefix flags of the token table (see QvTables::tok_pfx)
        const uint16_t *tk = (const uint16_t *)blob.get("tok");
        std::vector<uint8_t> pfx(N, 0);
        for (int v = 0; v < N; ++v)
            for (int k = 1; k < QV_MAX_SPAN; ++k) {
                const uint32_t a0 = to[(size_t)v * QV_MAX_SPAN + k - 1], a1 = to[(size_t)v * QV_MAX_SPAN + k], b1 = to[(size_t)v * QV_MAX_SPAN + k + 1];
                const uint32_t la = a1 - a0, lb = b1 - a1;
                if (la > 0 && lb >= la && memcmp(tk + a0, tk + a1, sizeof(uint16_t) * la) == 0) pfx[v] |= (uint8_t)(1u << (k - 1));
            }
        QV_TRY(upload(eng, pfx.data(), pfx.size(), &t.tok_pfx));
    }
    const uint32_t *po = (const uint32_t *)blob.get("piece_off");
    QV_TRY(upload(eng, po, (size_t)QV_VOCAB + 1, &t.piece_off));
    QV_TRY(upload(eng, (const uint8_t *)blob.get("piece_codes"), (size_t)po[QV_VOCAB] + 16, &t.piece_codes));
    return QV_OK;
}

// make context k the current one: the flat engine fields (and the model's activation pointers)
// are what every launcher reads
static void qv_select_ctx(qv_engine *eng, int k) {
    QvCtx &old = eng->ctx[eng->cur_ctx];
    old.last_batch = eng->last_batch;
    old.last_tmax = eng->last_tmax;
    QvCtx &c = eng->ctx[k];
    eng->work = c.work;
    eng->logprobs_ws = c.logprobs_ws;
    eng->t_host_scratch = c.t_host_scratch;
    eng->t_dev = c.t_dev;
    eng->last_batch = c.last_batch;
    eng->last_tmax = c.last_tmax;
    eng->cur_ctx = k;
    if (eng->model) qv_model_select_ctx(eng->model, k);
}

static int alloc_work(qv_engine *eng, int k) {
    QvWork &w = eng->work;
    int B = eng->cfg.max_batch, N = eng->tab.n_verses;
    w.max_batch = B;
    w.t_cap = qv_frames_for_samples(eng->cfg.max_samples) + 2;
    size_t Bz = (size_t)B;
    QV_TRY(dalloc(eng, Bz, &w.utt));
    QV_TRY(dalloc(eng, Bz * w.t_cap, &w.frame_ids));
    QV_TRY(dalloc(eng, Bz * w.t_cap, &w.greedy));
    QV_TRY(dalloc(eng, Bz * QV_MAXQ + 64, &w.q));
    QV_TRY(dalloc(eng, Bz * QV_MAXQ + 64, &w.qs));
    QV_TRY(dalloc(eng, Bz * 2 * QV_NSYM * QV_MAXW, &w.pm));
    QV_TRY(dalloc(eng, Bz * N, &w.cand1));
    QV_TRY(dalloc(eng, Bz * N * 3, &w.lcsf));
    QV_TRY(dalloc(eng, Bz * N * 3, &w.fs));
    QV_TRY(dalloc(eng, Bz * N, &w.lcs_p3));
    QV_TRY(dalloc(eng, Bz * N * 3, &w.frag_list));
    QV_TRY(dalloc(eng, (size_t)4, &w.frag_ctr));
    w.frag_cap = (int)(Bz * N * 3);
    w.search_sc = nullptr;
    QV_TRY(dalloc(eng, Bz * QV_RUNNER_CAP, &w.runner_idx));
    QV_TRY(dalloc(eng, Bz * QV_RUNNER_CAP, &w.runner_score));
    QV_TRY(dalloc(eng, Bz * QV_RUNNER_CAP, &w.top_search));
    QV_TRY(dalloc(eng, Bz * QV_RUNNER_CAP, &w.top_search_sc));
    QV_TRY(dalloc(eng, Bz * QV_RUNNER_CAP, &w.top_p3));
    QV_TRY(dalloc(eng, Bz * QV_RUNNER_CAP, &w.top_p3_sc));
    QV_TRY(dalloc(eng, Bz * QV_SPAN_BLOCKS, &w.span_part_score));
    QV_TRY(dalloc(eng, Bz * QV_SPAN_BLOCKS, &w.span_part_key));
    QV_TRY(dalloc(eng, Bz * QV_CAND_CAP, &w.cand_start));
    QV_TRY(dalloc(eng, Bz * QV_CAND_CAP, &w.cand_span));
    QV_TRY(dalloc(eng, Bz * QV_CAND_CAP, &w.cand_score));
    QV_TRY(dalloc(eng, Bz * QV_CAND_CAP, &w.cand_lead));
    QV_TRY(dalloc(eng, Bz * QV_CAND_CAP * QV_MAX_SPAN, &w.cand_memb));
    QV_TRY(dalloc(eng, Bz * QV_CAND_CAP, &w.cand_loss));
    QV_TRY(dalloc(eng, Bz * QV_CAND_CAP, &w.cand_final));
    QV_TRY(dalloc(eng, Bz, &w.results));
    QV_TRY(dalloc(eng, Bz * 4, &w.packed));
    QV_TRY(dalloc(eng, Bz, &w.fail_list));
    QV_TRY(dalloc(eng, (size_t)1, &w.n_fail));
    QV_TRY(dalloc(eng, Bz, &eng->t_dev));
    QV_HIP(hipHostMalloc((void **)&eng->t_host_scratch, sizeof(int32_t) * Bz * QV_STAGE_SLOTS, hipHostMallocDefault));
    eng->ctx[k].t_host_scratch = eng->t_host_scratch;  // owned by the context from here on
    eng->logprobs_ws = nullptr;
    if (eng->cfg.with_model) QV_TRY(dalloc(eng, Bz * w.t_cap * QV_VOCAB, &eng->logprobs_ws));
    QvCtx &c = eng->ctx[k];
    c.work = w;
    c.logprobs_ws = eng->logprobs_ws;
    c.t_host_scratch = eng->t_host_scratch;
    c.t_dev = eng->t_dev;
    c.busy = false;
    c.last_batch = c.last_tmax = 0;
    for (int i = 0; i < QV_STAGE_SLOTS; ++i) {
        QV_HIP(hipEventCreateWithFlags(&c.t_copied[i], hipEventDisableTiming));
        c.t_pending[i] = false;
    }
    c.t_slot = 0;
    if (eng->n_ctx > 1) {
        // QVERSE_CU_PARTITION=1 (experiment, DESIGN.md "Batches in flight"): context k's stream only sees its share of
        // the CUs of EVERY XCD (mask bit c * 8 + x = CU c of XCD x, tools/cumask_probe.hip; an XCD with an empty mask
        // would be unrestricted).  Masked streams are BLOCKING streams: the caller must not use the legacy default
        // stream for its own work or the contexts serialise behind it.  Four 64-CU partitions with 256 x 256 tiles
        // everywhere reach the same throughput as three unpartitioned batches in flight (16.6 k vs 16.6 k utt/s).
        static const int part = [] { const char *e = getenv("QVERSE_CU_PARTITION"); return e ? atoi(e) : 0; }();
        if (part) {
            uint32_t mask[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            const int lo = 32 * k / eng->n_ctx, hi = 32 * (k + 1) / eng->n_ctx;
            for (int cu = lo; cu < hi; ++cu)
                for (int x = 0; x < 8; ++x) mask[(cu * 8 + x) >> 5] |= 1u << ((cu * 8 + x) & 31);
            QV_HIP(hipExtStreamCreateWithCUMask(&c.stream, 8, mask));
        } else
        QV_HIP(hipStreamCreateWithFlags(&c.stream, hipStreamNonBlocking));
        QV_HIP(hipEventCreateWithFlags(&c.in_ready, hipEventDisableTiming));
        QV_HIP(hipEventCreateWithFlags(&c.done, hipEventDisableTiming));
    }
    return QV_OK;
}

extern "C" int qv_create(const qv_config *cfg, qv_engine **out) {
    if (!cfg || !out || cfg->struct_size != (int32_t)sizeof(qv_config)) {
        qv_set_error(nullptr, "qv_create: bad config (struct_size mismatch?)");
        return QV_ERR_ARG;
    }
    qv_engine *eng = new qv_engine();
    eng->cfg = *cfg;
    eng->model = nullptr;
    eng->t_host_scratch = nullptr;
    eng->last_batch = eng->last_tmax = 0;
    eng->n_ctx = cfg->n_contexts < 1 ? 1 : cfg->n_contexts;
    eng->cur_ctx = eng->next_ctx = 0;
    eng->profile_stages = false;
    eng->inject_lp = nullptr;
    eng->inject_tmax = eng->inject_batch = 0;
    for (QvCtx &c : eng->ctx) {
        c = QvCtx();
        c.t_host_scratch = nullptr;
        c.stream = nullptr;
        c.in_ready = c.done = nullptr;
        for (hipEvent_t &e : c.t_copied) e = nullptr;
        for (hipEvent_t &e : c.stage_ev) e = nullptr;
        c.stage_valid = false;
        c.n_post_graph = 0;
    }
    auto fail = [&](int rc) {
        { std::lock_guard<std::mutex> lk(g_create_error_mu); g_create_error = eng->last_error; }
        qv_destroy(eng);
        return rc;
    };
    if (cfg->max_span < 2 || cfg->max_span > QV_MAX_SPAN) { qv_set_error(eng, "CTC_DIRECT_MAX_SPAN must be in [2,6]"); return fail(QV_ERR_ARG); }
    // c2c-direct/run.py:67 takes any float (negative weights included); only a non-finite one is refused here: inf * 0.0
    // text scores would put NaNs into the final scores and leave the winner undefined
    if (!std::isfinite(cfg->text_weight)) { qv_set_error(eng, "CTC_DIRECT_TEXT_WEIGHT must be a finite number"); return fail(QV_ERR_ARG); }
    if (cfg->top_text < 1 || cfg->top_text > QV_RUNNER_CAP - 1) { qv_set_error(eng, "CTC_DIRECT_TOP_TEXT must be in [1,127]"); return fail(QV_ERR_ARG); }
    if (cfg->top_span_refs < 0 || cfg->top_span_refs > 128) { qv_set_error(eng, "CTC_DIRECT_TOP_SPAN_REFS must be in [0,128]"); return fail(QV_ERR_ARG); }
    if (cfg->max_batch < 1 || cfg->max_samples < 400) { qv_set_error(eng, "bad capacity"); return fail(QV_ERR_ARG); }
    // the CTC rerank keeps a candidate's 2L+1 <= T states in one wave's registers: T <= 768 frames.  The
    // reference has no such limit (c2c-direct/run.py:332 only asks 2L+1 <= T); refuse the capacity instead
    // of silently dropping long candidates.
    if (qv_frames_for_samples(cfg->max_samples) + 2 > 768) {
        qv_set_error(eng, "max_samples above 976,000 (61 s) is not supported: the CTC rerank handles at most 768 encoder frames");
        return fail(QV_ERR_CAPACITY);
    }
    if (cfg->n_contexts > QV_MAX_CTX) { qv_set_error(eng, "n_contexts must be in [1,8]"); return fail(QV_ERR_ARG); }
    eng->knobs = {cfg->top_text, cfg->top_span_refs, cfg->max_span, cfg->threshold, cfg->text_weight,
                  cfg->span_penalty, cfg->skip_unused_passes};
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= cfg->device) {
        qv_set_error(eng, "no HIP device available (libqverse needs a gfx950 GPU; there is no CPU fallback)");
        return fail(QV_ERR_HIP);
    }
    if (hipSetDevice(cfg->device) != hipSuccess) { qv_set_error(eng, "hipSetDevice failed"); return fail(QV_ERR_HIP); }
    eng->device = cfg->device;
    // The legacy default stream must own its hardware queue BEFORE the engine creates streams of its own.  A caller that
    // enqueues from the default stream (torch's current stream unless told otherwise) has qv_predict_batch_async record
    // an event there per batch; in a process whose first device work is this function, the runtime used to hand the
    // default stream -- first used later, by the table upload below -- a queue one of the context streams ends up on as
    // well, so that event sat behind ~300 queued kernels of an older batch and every new batch waited for it:
    // 4.7-4.8 ms per batch of 64 x 10 s instead of 3.5 (profiles/archive/r05_r_init_order.log; tools/sweep.py, which creates
    // its engine first, under-reported every row since round 2).  One kernel on the default stream, before the probe
    // below creates the process's first other streams, is what a process that touched the device earlier had anyway.
    // (A host thread that is inside a stream capture must not touch the legacy stream -- it would invalidate the capture or
    // be illegal in global mode -- so the warm-up is skipped then; such a host has used the device already.  The wait blocks
    // on whatever the host queued on blocking streams before: INTEGRATION.md "Creating the engine".)
    {
        hipStreamCaptureMode mode = hipStreamCaptureModeRelaxed;
        const bool capturing_ok = hipThreadExchangeStreamCaptureMode(&mode) == hipSuccess;   // relaxed for the probe below
        hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
        const bool legacy_capturing = hipStreamIsCapturing(nullptr, &st) == hipSuccess && st != hipStreamCaptureStatusNone;
        if (capturing_ok) (void)hipThreadExchangeStreamCaptureMode(&mode);                    // restore the host's mode
        (void)hipGetLastError();
        if (!legacy_capturing) {
            hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, nullptr, 1LL);
            if (hipStreamSynchronize(nullptr) != hipSuccess) { qv_set_error(eng, "default-stream warm-up kernel failed"); return fail(QV_ERR_HIP); }
        }
    }
    // four and more batches in flight only pay with a hardware queue per context stream (and one for the caller); when
    // the runtime runs fewer streams side by side than that -- GPU_MAX_HW_QUEUES unset or set after HIP initialised --
    // three contexts is the best measured setting (17.0 k utt/s on 4 or 8 queues; four contexts on 4 queues: 14.8 k)
    if (eng->n_ctx >= 4 && qv_probe_concurrent_streams() < QV_PROBE_STREAMS) eng->n_ctx = 3;
    int rc = load_tables(eng, cfg->tables_path);
    if (rc) return fail(rc);
    for (int k = 0; k < eng->n_ctx; ++k) {
        rc = alloc_work(eng, k);
        if (rc) return fail(rc);
    }
    {   // verse tracker workspace (qv_tracker_match)
        QvTrack &t = eng->track;
        if ((rc = dalloc(eng, (size_t)QV_TRACK_CAP * QV_MAXQ, &t.q)) || (rc = dalloc(eng, (size_t)QV_TRACK_CAP * 4, &t.meta)) ||
            (rc = dalloc(eng, (size_t)QV_TRACK_CAP * QV_TRACK_BLOCKS, &t.part_s)) ||
            (rc = dalloc(eng, (size_t)QV_TRACK_CAP * QV_TRACK_BLOCKS, &t.part_k)) ||
            (rc = dalloc(eng, (size_t)QV_TRACK_CAP, &t.out)))
            return fail(rc);
    }
    if (cfg->with_model) {
        rc = qv_model_create(eng, cfg, &eng->model);
        if (rc) return fail(rc);
    }
    qv_select_ctx(eng, 0);
    *out = eng;
    return QV_OK;
}

extern "C" void qv_destroy(qv_engine *e) {
    if (e && e->tail_ev) { (void)hipEventDestroy(e->tail_ev); e->tail_ev = nullptr; }
    if (!e) return;
    (void)hipDeviceSynchronize();
    if (e->model) qv_model_destroy(e->model);
    for (void *p : e->allocs) (void)hipFree(p);
    for (QvCtx &c : e->ctx) {
        if (c.t_host_scratch) (void)hipHostFree(c.t_host_scratch);
        if (c.stream) (void)hipStreamDestroy(c.stream);
        if (c.in_ready) (void)hipEventDestroy(c.in_ready);
        if (c.done) (void)hipEventDestroy(c.done);
        for (hipEvent_t e2 : c.t_copied) if (e2) (void)hipEventDestroy(e2);
        for (hipEvent_t e2 : c.stage_ev) if (e2) (void)hipEventDestroy(e2);
        for (int i = 0; i < c.n_post_graph; ++i) (void)hipGraphExecDestroy(c.post_graph[i].exec);
    }
    delete e;
}

extern "C" int qv_forward(qv_engine *eng, const float *audio_dev, const int64_t *lengths_host, int32_t batch,
                          int64_t n_max, float *logprobs_dev, int32_t t_max, int32_t *t_out_host, void *stream) {
    QV_SERIALISE(eng);
    QV_ORDERED(eng, stream);
    if (!eng) return QV_ERR_ARG;
    if (!eng->model) { qv_set_error(eng, "engine created without a model (with_model = 0)"); return QV_ERR_NO_MODEL; }
    return qv_model_forward(eng, eng->model, audio_dev, lengths_host, batch, n_max, logprobs_dev, t_max, t_out_host,
                            (hipStream_t)stream, /*zero_pad_rows=*/true);
}

extern "C" int qv_decode_retrieve_rerank_async(qv_engine *eng, const float *lp, const int32_t *t_host, int32_t batch,
                                               int32_t t_max, void *stream) {
    QV_SERIALISE(eng);
    QV_ORDERED(eng, stream);
    if (!eng || !lp || !t_host || batch < 1) return QV_ERR_ARG;
    qv_stage_mark(eng, 0, (hipStream_t)stream);   // no forward in this call: forward = 0
    qv_stage_mark(eng, 1, (hipStream_t)stream);
    return qv_post_run(eng, lp, t_max, t_host, batch, (hipStream_t)stream);
}

void qv_stage_mark(qv_engine *eng, int i, hipStream_t s) {
    if (!eng->profile_stages) return;
    QvCtx &c = eng->ctx[eng->cur_ctx];
    if (!c.stage_ev[i]) return;
    if (i == 0) c.stage_valid = false;
    if (hipEventRecord(c.stage_ev[i], s) == hipSuccess && i == 4) c.stage_valid = true;
}

extern "C" int qv_profile_inject_logprobs(qv_engine *eng, const float *logprobs_dev, int32_t t_max, const int32_t *t_host, int32_t batch) {
    QV_SERIALISE(eng);
    if (!eng) return QV_ERR_ARG;
    if (!logprobs_dev) { eng->inject_lp = nullptr; eng->inject_batch = 0; return QV_OK; }
    if (!t_host || batch < 1 || t_max < 1 || t_max > eng->work.t_cap) { qv_set_error(eng, "qv_profile_inject_logprobs: bad shape"); return QV_ERR_ARG; }
    eng->inject_t.assign(t_host, t_host + batch);
    eng->inject_lp = logprobs_dev;
    eng->inject_tmax = t_max;
    eng->inject_batch = batch;
    return QV_OK;
}

extern "C" int qv_profile_stages(qv_engine *eng, int32_t enable) {
    QV_SERIALISE(eng);
    if (!eng) return QV_ERR_ARG;
    if (enable)
        for (int k = 0; k < eng->n_ctx; ++k)
            for (hipEvent_t &e : eng->ctx[k].stage_ev)
                if (!e) QV_HIP(hipEventCreate(&e));
    eng->profile_stages = enable != 0;
    return QV_OK;
}

extern "C" int qv_stage_times(qv_engine *eng, int32_t k, float *ms4) {
    QV_SERIALISE(eng);
    if (!eng || !ms4 || k < 0 || k >= eng->n_ctx) return QV_ERR_ARG;
    QvCtx &c = eng->ctx[k];
    if (!c.stage_valid) { qv_set_error(eng, "qv_stage_times: no profiled batch on this context (qv_profile_stages first)"); return QV_ERR_ARG; }
    QV_HIP(hipEventSynchronize(c.stage_ev[4]));
    for (int i = 0; i < 4; ++i) QV_HIP(hipEventElapsedTime(&ms4[i], c.stage_ev[i], c.stage_ev[i + 1]));
    return QV_OK;
}

extern "C" int qv_fetch_results(qv_engine *eng, int32_t batch, int32_t t_max, qv_result *res, int32_t *greedy_host,
                                void *stream_) {
    QV_SERIALISE(eng);
    QV_ORDERED(eng, stream_);
    hipStream_t stream = (hipStream_t)stream_;
    if (!eng || !res || batch < 1 || batch > eng->work.max_batch) return QV_ERR_ARG;
    QV_HIP(hipMemcpyAsync(res, eng->work.results, sizeof(qv_result) * batch, hipMemcpyDeviceToHost, stream));
    if (greedy_host) {
        int tc = eng->work.t_cap;
        int w = t_max < tc ? t_max : tc;
        QV_HIP(hipMemcpy2DAsync(greedy_host, sizeof(int32_t) * t_max, eng->work.greedy, sizeof(int32_t) * tc,
                                sizeof(int32_t) * w, batch, hipMemcpyDeviceToHost, stream));
    }
    QV_HIP(hipStreamSynchronize(stream));
    for (int b = 0; b < batch; ++b) {
        // authoritative score of a CTC winner: math.exp(-ctc_norm_loss) in host double libm,
        // exactly what mixed/run.py:103-107 evaluates
        if (res[b].source == QV_SOURCE_CTC) res[b].score = exp(-(double)res[b].ctc_norm_loss);
        if (res[b].flags & QV_FLAG_TRANSCRIPT_TRUNCATED) { res[b].surah = res[b].ayah = res[b].ayah_end = 0; }
    }
    return QV_OK;
}

extern "C" int qv_decode_retrieve_rerank(qv_engine *eng, const float *lp, const int32_t *t_host, int32_t batch,
                                         int32_t t_max, qv_result *res, int32_t *greedy_host, void *stream) {
    QV_SERIALISE(eng);
    int rc = qv_decode_retrieve_rerank_async(eng, lp, t_host, batch, t_max, stream);
    if (rc) return rc;
    return qv_fetch_results(eng, batch, t_max, res, greedy_host, stream);
}

extern "C" int qv_predict_batch_async(qv_engine *eng, const float *audio_dev, const int64_t *lengths_host,
                                      int32_t batch, int64_t n_max, void *stream) {
    QV_SERIALISE(eng);
    QV_ORDERED(eng, stream);
    if (!eng) return QV_ERR_ARG;
    if (!eng->model) { qv_set_error(eng, "engine created without a model (with_model = 0)"); return QV_ERR_NO_MODEL; }
    if (batch > eng->work.max_batch) { qv_set_error(eng, "batch exceeds engine capacity"); return QV_ERR_CAPACITY; }
    int64_t lmax = 0;
    for (int b = 0; b < batch; ++b) lmax = std::max(lmax, lengths_host[b]);
    int t_max = qv_frames_for_samples(lmax);
    if (t_max > eng->work.t_cap) { qv_set_error(eng, "audio longer than engine capacity"); return QV_ERR_CAPACITY; }
    std::vector<int32_t> t_out(batch);
    hipStream_t run = (hipStream_t)stream;
    if (eng->n_ctx > 1) {
        // rotate to the next context; the host blocks only if that context's previous batch is
        // still in flight (bounds the queue and protects its pinned staging buffers)
        int k = eng->next_ctx;
        eng->next_ctx = (k + 1) % eng->n_ctx;
        QvCtx &c = eng->ctx[k];
        if (c.busy) QV_HIP(hipEventSynchronize(c.done));
        qv_select_ctx(eng, k);
        QV_HIP(hipEventRecord(c.in_ready, (hipStream_t)stream));
        QV_HIP(hipStreamWaitEvent(c.stream, c.in_ready, 0));
        run = c.stream;
    }
    qv_stage_mark(eng, 0, run);
    int rc = qv_model_forward(eng, eng->model, audio_dev, lengths_host, batch, n_max, eng->logprobs_ws, t_max,
                              t_out.data(), run, /*zero_pad_rows=*/false, /*may_graph=*/eng->n_ctx > 1);
    if (rc) return rc;
    qv_stage_mark(eng, 1, run);
    if (eng->inject_lp) {
        if (batch > eng->inject_batch) { qv_set_error(eng, "injected log-probs hold fewer utterances than the batch"); return QV_ERR_ARG; }
        rc = qv_post_run(eng, eng->inject_lp, eng->inject_tmax, eng->inject_t.data(), batch, run);
    } else
    rc = qv_post_run(eng, eng->logprobs_ws, t_max, t_out.data(), batch, run);
    if (rc) return rc;
    if (eng->n_ctx > 1) {
        QvCtx &c = eng->ctx[eng->cur_ctx];
        QV_HIP(hipEventRecord(c.done, c.stream));
        c.busy = true;
    }
    return QV_OK;
}

extern "C" int qv_predict_batch_async_ctx(qv_engine *eng, const float *audio_dev, const int64_t *lengths_host, int32_t batch,
                                          int64_t n_max, void *stream, int32_t *ctx_out) {
    QV_SERIALISE(eng);   // (recursive: the call below takes it again) -- the context id is read under the same lock hold
    int rc = qv_predict_batch_async(eng, audio_dev, lengths_host, batch, n_max, stream);
    if (rc == QV_OK && ctx_out) *ctx_out = eng->cur_ctx;
    return rc;
}

extern "C" int qv_predict_batch(qv_engine *eng, const float *audio_dev, const int64_t *lengths_host, int32_t batch,
                                int64_t n_max, qv_result *res, int32_t *greedy_host, void *stream) {
    QV_SERIALISE(eng);
    int rc = qv_predict_batch_async(eng, audio_dev, lengths_host, batch, n_max, stream);
    if (rc) return rc;
    if (eng->n_ctx > 1) return qv_fetch_results_ctx(eng, eng->cur_ctx, batch, eng->last_tmax, res, greedy_host);
    return qv_fetch_results(eng, batch, eng->last_tmax, res, greedy_host, stream);
}

static int fir_for(qv_engine *eng, const float *taps, int32_t n_taps, int32_t up, const qv_engine::Fir **out);

extern "C" int qv_upfirdn(qv_engine *eng, const float *x_dev, int64_t n_in, const float *taps, int32_t n_taps, int32_t up,
                          int32_t down, int64_t m0, int64_t n_out, float *y_dev, void *stream) {
    QV_SERIALISE(eng);
    QV_ORDERED(eng, stream);
    if (!eng || !x_dev || !taps || !y_dev || n_in < 1 || n_taps < 1 || up < 1 || down < 1 || m0 < 0 || n_out < 0) return QV_ERR_ARG;
    const qv_engine::Fir *fir = nullptr;
    QV_TRY(fir_for(eng, taps, n_taps, up, &fir));
    launch_upfirdn(x_dev, n_in, fir->hflip_dev, fir->P, up, down, m0, n_out, y_dev, (hipStream_t)stream);
    QV_HIP(hipGetLastError());
    return QV_OK;
}

// the per-phase, time-reversed filter rows of (taps, up), uploaded once per distinct filter
static int fir_for(qv_engine *eng, const float *taps, int32_t n_taps, int32_t up, const qv_engine::Fir **out) {
    for (const auto &f : eng->firs)
        if (f.up == up && (int)f.taps.size() == n_taps && memcmp(f.taps.data(), taps, sizeof(float) * n_taps) == 0) { *out = &f; return QV_OK; }
    int P = (n_taps + up - 1) / up;
    std::vector<float> hf((size_t)up * P, 0.f);
    for (int t = 0; t < up; ++t)
        for (int j = 0; j < P; ++j) {
            int k = t + up * (P - 1 - j);
            if (k < n_taps) hf[(size_t)t * P + j] = taps[k];
        }
    float *d = nullptr;
    QV_TRY(dalloc(eng, hf.size(), &d));
    QV_HIP(hipMemcpy(d, hf.data(), hf.size() * sizeof(float), hipMemcpyHostToDevice));
    eng->firs.push_back({up, std::vector<float>(taps, taps + n_taps), d, P});
    *out = &eng->firs.back();
    return QV_OK;
}

extern "C" int qv_upfirdn_batch(qv_engine *eng, const float *x_dev, int64_t x_pitch, const int32_t *src_rows_host,
                                const int64_t *n_in_host, int32_t rows, const float *taps, int32_t n_taps, int32_t up, int32_t down,
                                int64_t m0, float *y_dev, int64_t y_pitch, void *stream) {
    QV_SERIALISE(eng);
    QV_ORDERED(eng, stream);
    if (!eng || !x_dev || !n_in_host || !taps || !y_dev || rows < 1 || rows > QV_RESAMPLE_ROWS || n_taps < 1 || up < 1 || down < 1 || m0 < 0 ||
        x_pitch < 1 || y_pitch < 1)
        return QV_ERR_ARG;
    for (int r = 0; r < rows; ++r) {
        if (n_in_host[r] < 1 || n_in_host[r] > x_pitch) { qv_set_error(eng, "qv_upfirdn_batch: a row's length is outside [1, x_pitch]"); return QV_ERR_ARG; }
        if ((n_in_host[r] * up + down - 1) / down > y_pitch) { qv_set_error(eng, "qv_upfirdn_batch: y_pitch is shorter than a row's output"); return QV_ERR_ARG; }
        if (src_rows_host && src_rows_host[r] < 0) return QV_ERR_ARG;
    }
    const qv_engine::Fir *fir = nullptr;
    QV_TRY(fir_for(eng, taps, n_taps, up, &fir));
    // row table: [QV_RESAMPLE_ROWS] int64 lengths + int32 source rows, one slot of a small ring per call (the copy is issued
    // from pageable memory: the runtime stages it before this function returns, so the host vectors may go away)
    if (!eng->resample_tab) QV_TRY(dalloc(eng, (size_t)QV_RESAMPLE_SLOTS * QV_RESAMPLE_ROWS * 12, &eng->resample_tab));
    unsigned char *slot = eng->resample_tab + (size_t)(eng->resample_at++ % QV_RESAMPLE_SLOTS) * QV_RESAMPLE_ROWS * 12;
    hipStream_t s = (hipStream_t)stream;
    QV_HIP(hipMemcpyAsync(slot, n_in_host, sizeof(int64_t) * rows, hipMemcpyHostToDevice, s));
    int32_t *src_dev = nullptr;
    if (src_rows_host) {
        src_dev = (int32_t *)(slot + (size_t)QV_RESAMPLE_ROWS * 8);
        QV_HIP(hipMemcpyAsync(src_dev, src_rows_host, sizeof(int32_t) * rows, hipMemcpyHostToDevice, s));
    }
    launch_upfirdn_rows(x_dev, x_pitch, src_dev, (const int64_t *)slot, rows, fir->hflip_dev, fir->P, up, down, m0, y_dev, y_pitch, s);
    QV_HIP(hipGetLastError());
    return QV_OK;
}

extern "C" int qv_mixdown_batch(qv_engine *eng, const float *x_dev, int64_t x_pitch, const int64_t *n_frames_host, int32_t rows,
                                int32_t channels, float *y_dev, int64_t y_pitch, void *stream) {
    QV_SERIALISE(eng);
    QV_ORDERED(eng, stream);
    if (!eng || !x_dev || !n_frames_host || !y_dev || rows < 1 || rows > QV_RESAMPLE_ROWS || channels < 1 || channels > 64) return QV_ERR_ARG;
    int64_t mx = 0;
    for (int r = 0; r < rows; ++r) {
        if (n_frames_host[r] < 0 || n_frames_host[r] * channels > x_pitch || n_frames_host[r] > y_pitch) return QV_ERR_ARG;
        mx = n_frames_host[r] > mx ? n_frames_host[r] : mx;
    }
    if (!eng->resample_tab) QV_TRY(dalloc(eng, (size_t)QV_RESAMPLE_SLOTS * QV_RESAMPLE_ROWS * 12, &eng->resample_tab));
    unsigned char *slot = eng->resample_tab + (size_t)(eng->resample_at++ % QV_RESAMPLE_SLOTS) * QV_RESAMPLE_ROWS * 12;
    hipStream_t s = (hipStream_t)stream;
    QV_HIP(hipMemcpyAsync(slot, n_frames_host, sizeof(int64_t) * rows, hipMemcpyHostToDevice, s));
    launch_mixdown(x_dev, x_pitch, (const int64_t *)slot, rows, channels, y_dev, y_pitch, mx, s);
    QV_HIP(hipGetLastError());
    return QV_OK;
}

extern "C" const int32_t *qv_packed_results_dev(qv_engine *eng) { return eng ? eng->work.packed : nullptr; }

extern "C" int32_t qv_context_count(const qv_engine *eng) { return eng ? eng->n_ctx : 0; }   // fixed at qv_create
extern "C" int32_t qv_last_context(const qv_engine *eng) {
    if (!eng) return -1;
    std::lock_guard<std::recursive_mutex> lk(const_cast<qv_engine *>(eng)->mu);
    return eng->cur_ctx;
}

extern "C" int qv_wait_ctx(qv_engine *eng, int32_t k) {
    QV_SERIALISE(eng);
    if (!eng || k < 0 || k >= eng->n_ctx) return QV_ERR_ARG;
    QvCtx &c = eng->ctx[k];
    if (eng->n_ctx > 1) {
        if (c.busy) QV_HIP(hipEventSynchronize(c.done));
    } else {
        QV_HIP(hipDeviceSynchronize());   // single-context engines run on the caller's stream, which we were not given
    }
    return QV_OK;
}

extern "C" const int32_t *qv_packed_results_ctx(qv_engine *eng, int32_t k, void *stream) {
    QV_SERIALISE(eng);
    if (!eng || k < 0 || k >= eng->n_ctx) return nullptr;
    QvCtx &c = eng->ctx[k];
    if (eng->n_ctx > 1 && c.busy && hipStreamWaitEvent((hipStream_t)stream, c.done, 0) != hipSuccess) return nullptr;
    return c.work.packed;
}

extern "C" int qv_fetch_results_ctx(qv_engine *eng, int32_t k, int32_t batch, int32_t t_max, qv_result *res,
                                    int32_t *greedy_host) {
    QV_SERIALISE(eng);
    if (!eng || k < 0 || k >= eng->n_ctx) return QV_ERR_ARG;
    int keep = eng->cur_ctx;
    if (eng->n_ctx == 1) QV_HIP(hipDeviceSynchronize());  // the batch ran on a caller stream we were not given
    qv_select_ctx(eng, k);
    // on the context's own stream, so the copy is ordered after its batch
    int rc = qv_fetch_results(eng, batch, t_max, res, greedy_host, eng->n_ctx > 1 ? eng->ctx[k].stream : nullptr);
    qv_select_ctx(eng, keep);
    return rc;
}

extern "C" int qv_debug_retrieve(qv_engine *eng, const uint8_t *codes_host, int32_t n_codes, int32_t *base_start,
                                 int32_t *base_span, double *base_score, int32_t *cand_start, int32_t *cand_span,
                                 double *cand_score, int32_t cand_cap, int32_t *n_cand, int32_t *runner_idx,
                                 double *runner_score, int32_t *n_runners, void *stream_) {
    QV_SERIALISE(eng);
    QV_ORDERED(eng, stream_);
    hipStream_t stream = (hipStream_t)stream_;
    if (!eng) return QV_ERR_ARG;
    int rc = qv_post_debug_retrieve(eng, codes_host, n_codes, stream);
    if (rc) return rc;
    QvUtt u;
    QV_HIP(hipMemcpy(&u, eng->work.utt, sizeof(QvUtt), hipMemcpyDeviceToHost));
    *base_start = u.base_start; *base_span = u.base_span; *base_score = u.base_score;
    int n = u.n_cand < cand_cap ? u.n_cand : cand_cap;
    *n_cand = u.n_cand;
    if (n > 0) {
        QV_HIP(hipMemcpy(cand_start, eng->work.cand_start, sizeof(int32_t) * n, hipMemcpyDeviceToHost));
        QV_HIP(hipMemcpy(cand_span, eng->work.cand_span, sizeof(int32_t) * n, hipMemcpyDeviceToHost));
        QV_HIP(hipMemcpy(cand_score, eng->work.cand_score, sizeof(double) * n, hipMemcpyDeviceToHost));
    }
    *n_runners = u.n_runners;
    if (u.n_runners > 0) {
        QV_HIP(hipMemcpy(runner_idx, eng->work.runner_idx, sizeof(int32_t) * u.n_runners, hipMemcpyDeviceToHost));
        QV_HIP(hipMemcpy(runner_score, eng->work.runner_score, sizeof(double) * u.n_runners, hipMemcpyDeviceToHost));
    }
    return QV_OK;
}

extern "C" int qv_tracker_match(qv_engine *eng, const uint8_t *codes_host, const int32_t *offsets_host,
                                const int32_t *n_words_host, const int32_t *bonus_verse_host, int32_t batch,
                                qv_track_match *out_host, void *stream) {
    QV_SERIALISE(eng);
    QV_ORDERED(eng, stream);
    if (!eng) return QV_ERR_ARG;
    if (batch < 0 || !offsets_host || !n_words_host || !bonus_verse_host || !out_host || (batch > 0 && !codes_host)) {
        qv_set_error(eng, "qv_tracker_match: null argument");
        return QV_ERR_ARG;
    }
    for (int b = 0; b < batch; ++b) {
        int n = offsets_host[b + 1] - offsets_host[b];
        if (n < 0 || bonus_verse_host[b] >= eng->tab.n_verses || n_words_host[b] < 0) {
            qv_set_error(eng, "qv_tracker_match: bad offsets / word count / bonus verse");
            return QV_ERR_ARG;
        }
        if (n > QV_MAXQ) { qv_set_error(eng, "qv_tracker_match: text longer than QV_MAX_TRANSCRIPT"); return QV_ERR_CAPACITY; }
    }
    return qv_post_tracker_match(eng, codes_host, offsets_host, n_words_host, bonus_verse_host, batch, out_host,
                                 (hipStream_t)stream);
}

// Entry points that run a single text through the CURRENT context's workspace on the caller's
// stream: with batches in flight, wait until no internal stream is still using a workspace.
static int quiesce_contexts(qv_engine *eng) {
    if (eng->n_ctx > 1)
        for (int k = 0; k < eng->n_ctx; ++k)
            if (eng->ctx[k].busy) QV_HIP(hipEventSynchronize(eng->ctx[k].done));
    return QV_OK;
}

extern "C" int qv_match_verse(qv_engine *eng, const uint8_t *codes_host, int32_t n_codes, int32_t n_bonus,
                              const int32_t *bonus_verse, const double *bonus_value, int32_t max_span, int32_t *start,
                              int32_t *span, double *score, void *stream) {
    QV_SERIALISE(eng);
    QV_ORDERED(eng, stream);
    if (!eng) return QV_ERR_ARG;
    if (n_codes < 0 || (n_codes > 0 && !codes_host) || n_bonus < 0 || n_bonus > 3 || (n_bonus > 0 && (!bonus_verse || !bonus_value)) ||
        max_span < 2 || max_span > 8 || !start || !span || !score) {
        qv_set_error(eng, "qv_match_verse: bad argument (n_bonus in [0,3], max_span in [2,8])");
        return QV_ERR_ARG;
    }
    for (int i = 0; i < n_bonus; ++i)
        if (bonus_verse[i] < 0 || bonus_verse[i] >= eng->tab.n_verses) { qv_set_error(eng, "qv_match_verse: bonus verse out of range"); return QV_ERR_ARG; }
    if (n_codes > QV_MAXQ) { qv_set_error(eng, "qv_match_verse: text longer than QV_MAX_TRANSCRIPT"); return QV_ERR_CAPACITY; }
    QV_TRY(quiesce_contexts(eng));
    int rc = qv_post_match_verse(eng, codes_host, n_codes, n_bonus, bonus_verse, bonus_value, max_span, (hipStream_t)stream);
    if (rc) return rc;
    QvUtt u;
    QV_HIP(hipMemcpy(&u, eng->work.utt, sizeof(QvUtt), hipMemcpyDeviceToHost));
    *start = u.base_start; *span = u.base_span; *score = u.base_score;
    return QV_OK;
}

extern "C" int qv_debug_ctc_loss(qv_engine *eng, const float *lp, int32_t T, const uint16_t *tg, const int32_t *lens,
                                 int32_t n, float *loss_host, void *stream) {
    QV_SERIALISE(eng);
    QV_ORDERED(eng, stream);
    if (!eng || n < 1) return QV_ERR_ARG;
    return qv_post_debug_ctc(eng, lp, T, tg, lens, n, loss_host, (hipStream_t)stream);
}

extern "C" int qv_debug_forward_tap(qv_engine *eng, int32_t what, int32_t layer, float *out_dev, void *stream) {
    QV_SERIALISE(eng);
    QV_ORDERED(eng, stream);
    if (!eng) return QV_ERR_ARG;
    if (!eng->model) { qv_set_error(eng, "engine created without a model"); return QV_ERR_NO_MODEL; }
    return qv_model_tap(eng, eng->model, what, layer, out_dev, (hipStream_t)stream);
}

// ---- measurement hooks -------------------------------------------------------------------
extern "C" int qv_profile_gemm(qv_engine *eng, int32_t enable) {
    if (!eng) return QV_ERR_ARG;
    qv_gemm_prof_enable(enable != 0);
    return QV_OK;
}

extern "C" int qv_profile_gemm_read(qv_engine *eng, double *ms21, double *flops21, int32_t *launches21) {
    if (!eng || !ms21 || !flops21 || !launches21) return QV_ERR_ARG;
    int n[21];
    qv_gemm_prof_collect(ms21, flops21, n);
    for (int i = 0; i < 21; ++i) launches21[i] = n[i];
    return QV_OK;
}

extern "C" int qv_profile_replay_kernel(qv_engine *eng, int32_t which, char *name_out, int32_t name_cap) {
    QV_SERIALISE(eng);
    if (!eng || !name_out || name_cap < 8) return QV_ERR_ARG;
    if (!eng->model) { qv_set_error(eng, "engine created without a model"); return QV_ERR_NO_MODEL; }
    return qv_model_replay_kernel(eng, eng->model, which, name_out, name_cap);
}

extern "C" int qv_weights_info(qv_engine *eng, char *out, int32_t cap) {
    QV_SERIALISE(eng);
    if (!eng || !out || cap < 16) return QV_ERR_ARG;
    if (!eng->model) { snprintf(out, (size_t)cap, "no acoustic model (post-logits stages only)"); return QV_OK; }
    qv_model_weights_info(eng->model, out, cap);
    return QV_OK;
}

extern "C" int qv_debug_attention_variant(int32_t mode) {
    if (mode < -1 || mode > 5) return QV_ERR_ARG;
    qv_attention_set_variant(mode);
    return QV_OK;
}

extern "C" int qv_debug_kernel_variant(int32_t which, int32_t mode) {
    if (which < 0 || which >= QV_KV_COUNT || mode < -1 || mode > 15) return QV_ERR_ARG;
    qv_kernel_variant_set(which, mode);
    return QV_OK;
}

extern "C" int qv_debug_forward_graph_stats(qv_engine *eng, int64_t *replays, int64_t *captures) {
    QV_SERIALISE(eng);
    if (!eng || !eng->model || !replays || !captures) return QV_ERR_ARG;
    qv_model_graph_stats(eng->model, replays, captures);
    return QV_OK;
}

extern "C" int64_t qv_debug_forward_graph_failures(qv_engine *eng) {
    QV_SERIALISE(eng);
    return (eng && eng->model) ? qv_model_graph_failures(eng->model) : -1;
}

extern "C" int qv_debug_gemm_tiles(int32_t mode) {
    if (mode < -1 || mode > 2) return QV_ERR_ARG;
    qv_gemm_set_t256(mode);
    return QV_OK;
}

extern "C" int qv_debug_gemm_tile_height(int32_t mode) {
    if (mode < -1 || mode > 3) return QV_ERR_ARG;
    qv_gemm_set_bm(mode);
    return QV_OK;
}

extern "C" int qv_profile_replay_gemm(qv_engine *eng, int32_t which, int32_t iters, double *avg_us, double *flops_per_launch,
                                      void *stream) {
    QV_SERIALISE(eng);
    if (!eng || !avg_us || !flops_per_launch) return QV_ERR_ARG;
    if (!eng->model) { qv_set_error(eng, "engine created without a model"); return QV_ERR_NO_MODEL; }
    return qv_model_replay_gemm(eng, eng->model, which, iters, avg_us, flops_per_launch, (hipStream_t)stream);
}

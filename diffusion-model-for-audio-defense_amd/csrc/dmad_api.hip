// libdmad_hip.so — C ABI (include/dmad.h) and host-side engine: weight packing into MFMA/LDS
// layouts, device workspace (allocated once in dmad_create), kernel sequencing on the caller's stream.
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <map>
#include <string>
#include <vector>

#include "../../include/dmad.h"
#include "dmad_common.h"
#include "unet_ops.h"
#include "elementwise.h"
#include "gemm_f32.h"
#include "gemm_h16.h"
#include "wn_bf16.h"

using namespace dmad;

namespace {

thread_local std::string g_err, g_warn;

// WaveNet paths of an engine (dmad_wavenet_eps_path / dmad_set_waveform_tier): the mode's default (16-bit where resident), exact fp32,
// the fp32 pipeline on split-f16 operands
enum { PATH_DEFAULT = 0, PATH_FP32 = 1, PATH_X3 = 2 };

int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define HIPCHK(x)                                                                                   \
    do {                                                                                            \
        hipError_t _e = (x);                                                                        \
        if (_e != hipSuccess) return fail(DMAD_ERR_HIP, "%s failed: %s (%s:%d)", #x, hipGetErrorString(_e), __FILE__, __LINE__); \
    } while (0)
#define CHK(x)                 \
    do {                       \
        int _r = (x);          \
        if (_r != 0) return _r; \
    } while (0)
// end of an entry point's launch sequence: a failed launch, or a GEMM argument block no kernel serves (gemm_f32.h)
#define LASTCHK()                                                                                   \
    do {                                                                                            \
        HIPCHK(hipGetLastError());                                                                  \
        if (int _b = gemm_take_bad_shapes() + gemm_h16_take_bad_shapes()) return fail(DMAD_ERR_INVALID, "%d GEMM launch(es) refused: unsupported shape", _b); \
    } while (0)

uint16_t f2bf(float f) {   // round-to-nearest-even, NaN stays NaN
    uint32_t u;
    memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);
    return (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}

uint16_t f2h(float f) {    // fp32 -> IEEE half, round-to-nearest-even (subnormals and overflow to inf included)
    uint32_t u;
    memcpy(&u, &f, 4);
    const uint32_t sign = (u >> 16) & 0x8000u, ax = u & 0x7fffffffu;
    if (ax > 0x7f800000u) return (uint16_t)(sign | 0x7e00u);                 // NaN
    if (ax >= 0x477ff000u) return (uint16_t)(sign | 0x7c00u);                // >= 65520 rounds to inf
    if (ax < 0x33000001u) return (uint16_t)sign;                             // <= 2^-25 rounds to zero
    int e = (int)(ax >> 23) - 127;
    uint32_t m = (ax & 0x7fffffu) | 0x800000u;                               // 24-bit significand
    int shift = e < -14 ? 13 + (-14 - e) : 13;                               // bits dropped (subnormal: more)
    uint32_t h = m >> shift, rem = m & ((1u << shift) - 1), half = 1u << (shift - 1);
    if (rem > half || (rem == half && (h & 1u))) ++h;
    if (e < -14) return (uint16_t)(sign | h);                                // subnormal (a carry into 0x400 is the smallest normal)
    return (uint16_t)(sign | (uint32_t)(((e + 15) << 10) + (h - 0x400u)));   // a significand carry bumps the exponent
}

// f2h with a census of what leaves the f16 normal range.  A Gaussian weight tensor always has a few values below 2^-14 (f16
// subnormals): their absolute rounding error (<= 2^-25) is far below the 2^-12 relative error of the tensor's typical weights
// and does not matter.  What does matter is a tensor (or a large part of one) that lives down there as a whole — a checkpoint
// with tiny weight-norm gains — or a value beyond 65504 (inf, NaN logits).  So the census is per packed tensor: the share of its
// squared norm carried by subnormal values; dmad_finalize_weights warns when that share exceeds 1e-6 in any tensor.
thread_local double g_h_sq = 0.0, g_h_sq_sub = 0.0;
thread_local long g_h_bad_tensors = 0, g_h_ovf = 0, g_h_tensors = 0;
thread_local double g_h_worst = 0.0;
uint16_t f2h_census(float f) {
    const float a = fabsf(f);
    g_h_sq += (double)a * a;
    if (a < 6.103515625e-05f) g_h_sq_sub += (double)a * a;
    if (a >= 65520.f) ++g_h_ovf;
    return f2h(f);
}
// Development switch of the low-toggle-weights experiment (tools/gpu_weight_toggle.py), compiled in ONLY with
// -DDMAD_DEV_WEIGHT_MASK (never in the product library: a leaked environment variable must not be able to void the exact-vote
// bounds): DMAD_WEIGHT_MASK_BITS = k rounds the f16 weight images of the 16-bit WaveNet path to 10 - k mantissa bits (round to nearest
// even on the f16 pattern, the k low bits then zero: fewer toggling bits on the L2 -> LDS -> register path at a precision cost);
// DMAD_WEIGHT_MASK_WHICH selects the images (bit 0 dilated conv, 1 res conv, 2 skip convs, 3 final_conv.0; default 3).
#ifdef DMAD_DEV_WEIGHT_MASK
thread_local int g_mask_bits = 0;
uint16_t f2h_census_masked(float f) {
    uint32_t h = f2h_census(f);
    const int k = g_mask_bits;
    if (k > 0 && (h & 0x7c00u) != 0x7c00u) {
        const uint32_t sign = h & 0x8000u;
        uint32_t m = h & 0x7fffu;
        m = (m + ((1u << (k - 1)) - 1u) + ((m >> k) & 1u)) & ~((1u << k) - 1u);      // a carry moves into the exponent as it should
        if (m > 0x7bffu) m = 0x7bffu & ~((1u << k) - 1u);
        h = sign | m;
    }
    return (uint16_t)h;
}
#endif
void census_close_tensor() {
    ++g_h_tensors;
    const double share = g_h_sq > 0.0 ? g_h_sq_sub / g_h_sq : 0.0;
    if (share > 1e-6) ++g_h_bad_tensors;
    if (share > g_h_worst) g_h_worst = share;
    g_h_sq = g_h_sq_sub = 0.0;
}

float h2f(uint16_t h) {    // IEEE half -> fp32 (exact)
    const uint32_t sign = (uint32_t)(h & 0x8000u) << 16, e = (h >> 10) & 0x1fu, m = h & 0x3ffu;
    uint32_t u;
    if (e == 0) {
        if (m == 0) u = sign;
        else {                                                               // subnormal: normalise
            int sh = 0;
            uint32_t mm = m;
            while (!(mm & 0x400u)) { mm <<= 1; ++sh; }
            u = sign | ((uint32_t)(127 - 15 - sh + 1) << 23) | ((mm & 0x3ffu) << 13);
        }
    } else if (e == 31) u = sign | 0x7f800000u | (m << 13);
    else u = sign | ((e + 112u) << 23) | (m << 13);
    float f;
    memcpy(&f, &u, 4);
    return f;
}

// fp32 matrix [rows][K] -> the split-f16 storage format of dmad_common.h (same bytes per row: every 4 consecutive k become
// the 16-byte chunk [hi0 hi1 hi2 hi3 lo0 lo1 lo2 lo3], lo = f16((x - hi) * 2^11))
void split_rows(const float* src, size_t count, float* dst) {
    uint16_t* o = (uint16_t*)dst;
    for (size_t i = 0; i < count; i += 4)
        for (int j = 0; j < 4; ++j) {
            const float x = src[i + j];
            const uint16_t hi = f2h(x);
            o[2 * i + j] = hi;
            o[2 * i + 4 + j] = f2h((x - h2f(hi)) * 2048.f);
        }
}

const int kVggCfg[] = {64, 64, -1, 128, 128, -1, 256, 256, 256, 256, -1, 512, 512, 512, 512, -1, 512, 512, 512, 512, -1};
const int kVggCfgLen = sizeof(kVggCfg) / sizeof(int);
constexpr int kMelLd = 1040;      // 1025 rFFT bins padded to a multiple of 16
constexpr int kDftM = 2050;       // 1025 cos rows + 1025 sin rows
constexpr int kDftLd = 2052;

struct HostW {
    std::vector<float> v;
    std::vector<int64_t> shape;
};

}  // namespace

struct dmad_engine {
    dmad_config cfg{};
    int L = 0, LP = 0, NL = 0, maxB = 0, LPm = 0;
    bool bf16 = true, f32 = false, wn_final = false, cls_final = false;   // bf16 / f32: which WaveNet paths are resident
    int maxB32 = 0;                        // clips per exact-fp32 WaveNet pass (== maxB for DMAD_FP32, recheck_batch for DMAD_EXACT)
    int mode = DMAD_MODE_FAST;             // enum dmad_mode (DMAD_EXACT engines switch at run time)
    int wave_tier = 2;                     // dmad_set_waveform_tier: WaveNet path of the waveform-returning entry points in DMAD_MODE_EXACT_VOTES
    float tau = 0.f;                       // recheck bound on the bf16 top-2 logit margin
    long long* rc_list = nullptr;          // global indices of the samples queued for the fp32 re-evaluation
    unsigned long long* rc_n = nullptr;    // their number (device) ...
    unsigned long long* rc_n_host = nullptr;   // ... and its pinned host mirror
    long rc_cap = 0;
    int64_t st_samples = 0, st_rechecked = 0;
    int diag[5] = {0, 0, 0, 0, 0};         // dmad_debug_rounding: GemmF32Args::diag of the x3 tier's dil / res / skip / f0 launches, init hi-only
    std::string warn;                      // dmad_last_warning
    std::map<std::string, HostW> hw;
    std::vector<void*> allocs;
    int64_t bytes = 0;
    int emb_t = -1;
    // optional per-launch timing of the dominant kernel (bench.py roofline): HIP event pairs on the launch stream
    bool prof_on = false;
    std::vector<hipEvent_t> prof_ev, prof_ev_f;      // layer launches / final-kernel launches
    size_t prof_used = 0, prof_used_f = 0;

    // WaveNet small fp32 params
    float *init_w = nullptr, *init_b = nullptr, *fc1w = nullptr, *fc1b = nullptr, *fc2w = nullptr, *fc2b = nullptr;
    float *fctw = nullptr, *fctb = nullptr, *emb_table = nullptr, *emb2 = nullptr, *epi_c = nullptr;
    float *bf0 = nullptr, *wz = nullptr;
    float bz = 0.f;
    // 16-bit MFMA path (operands bf16, or f16 when `f16` is set)
    bool f16 = false;
    h16_t *w1p = nullptr, *w2p = nullptr, *wsp = nullptr, *wf0p = nullptr;
    float *b1p = nullptr, *b2 = nullptr, *bskip_sum = nullptr;
    h16_t *hA = nullptr, *hB = nullptr, *gstore = nullptr;
    // fp32 path
    float *wdil = nullptr, *bdil = nullptr, *wrs = nullptr, *brs = nullptr, *wf0 = nullptr;
    float *wdil_x3 = nullptr, *wrs_x3 = nullptr, *wf0_x3 = nullptr;      // the same weights in the split-f16 storage format (x3 tier)
    float *wskip32 = nullptr, *wskip_x3 = nullptr, *bskip32 = nullptr;   // [NL][256][256] skip weights (fp32 / split-f16), sum of the skip biases
    float* gstore32 = nullptr;                                           // gate outputs of all layers [NL][maxB32 * L][256] (fp32 and x3 tiers)
    float tau2 = 0.f;                      // recheck bound of the x3 tier (its logit-difference error against the fp32 path)
    long long* rc_list2 = nullptr;         // samples the x3 tier leaves to the fp32 tier
    int64_t st_rechecked2 = 0;
    float *hA32 = nullptr, *hB32 = nullptr, *H32 = nullptr, *g32 = nullptr, *skip32 = nullptr;
    // common work buffers
    float *xt = nullptr, *eps = nullptr, *x0 = nullptr, *znoise = nullptr;
    // classifier
    float *dftA = nullptr, *fbA = nullptr, *mel_xp = nullptr, *dftD = nullptr, *melP = nullptr, *melM = nullptr, *spec = nullptr;
    float *vconv1w = nullptr;
    float* vconvw[16] = {nullptr};
    float* vscale[16] = {nullptr};
    float* vshift[16] = {nullptr};
    float* vfcw[3] = {nullptr};
    float* vfcb[3] = {nullptr};
    float *act0 = nullptr, *act1 = nullptr, *logits = nullptr, *slab = nullptr;
    long slab_floats = 0;
    // ResNeXt29 8x64d (models/resnext.py): 9 bottlenecks, every conv with its folded eval-BatchNorm scale/shift
    int cls_kind = 0;                      // 0 = VGG19_bn, 1 = ResNeXt29
    struct RxConv { float *w = nullptr, *scale = nullptr, *shift = nullptr; h16_t* wh = nullptr; float* wx = nullptr; };   // wh: f16 image with the BN scale folded in (16-bit tier); wx: split-f16 image (middle tier)
    struct RxBlock { RxConv reduce, conv, expand, shortc; bool has_short = false; int cin = 0, cout = 0, D = 0, stride = 1; };
    RxBlock rx[9];
    RxConv rxconv1;
    float *rxfcw = nullptr, *rxfcb = nullptr;
    float *rxX = nullptr, *rxY = nullptr, *rxT1 = nullptr, *rxT2 = nullptr, *rxS = nullptr;   // NHWC work buffers
    // ResNeXt29's 16-bit tier (engines with a 16-bit side): every conv through gemm_h16 on f16 operands (fp32 accumulate, BN shift /
    // shortcut add / ReLU in fp32), maps kept as f16 between the convs; tier 1 of the exact-vote loop uses it, the recheck tiers and
    // dmad_classify stay on the fp32 matrix cores
    bool rx_h16 = false;
    bool rx_x3 = false;                    // exact-vote engines: ResNeXt29's split-f16 tier (every conv as three f16 MFMAs per product: fp32-grade), tier 1 of the exact-vote loop
    h16_t *rxX16 = nullptr, *rxY16 = nullptr, *rxT1h = nullptr, *rxT2h = nullptr, *rxS16 = nullptr;
    // Improved-Diffusion UNet purifier on 1x32x32 mel spectrograms (improved_diffusion/unet.py:278-477)
    struct UnOp {                          // one module of a TimestepEmbedSequential
        int kind = 0;                      // 0 conv_in, 1 res, 2 attn, 3 down, 4 up
        int cin = 0, cout = 0;
        float *gn1w = nullptr, *gn1b = nullptr, *w1 = nullptr, *b1 = nullptr;       // res: in_layers; attn: norm, qkv; down/up/conv_in: conv
        float *embw = nullptr, *embb = nullptr, *gn2w = nullptr, *gn2b = nullptr, *w2 = nullptr, *b2 = nullptr;   // res: emb, out_layers; attn: proj_out
        float *skw = nullptr, *skb = nullptr;                                       // res: 1x1 skip_connection
        h16_t *w1h = nullptr, *w2h = nullptr, *skwh = nullptr;                      // f16 images of w1 / w2 / skw (16-bit tier)
        float *w1x = nullptr, *w2x = nullptr, *skwx = nullptr;                      // the same weights in the split-f16 storage format (middle tier)
        size_t ss_off = 0;                 // res: offset of its (scale, shift) row [2 * cout] inside a step's row of un_ss_table
    };
    std::vector<std::vector<UnOp>> un_in, un_out;
    std::vector<UnOp> un_mid;
    std::vector<int> un_hs_ch, un_hs_hw;   // channels / pixels of the saved input-block outputs
    std::vector<float*> un_hs;
    bool un_final = false;
    int un_t = -1;
    // Every ResBlock's emb_layers output depends on the step t alone (unet.py:186-199), and a sampler walks the same few steps for
    // every batch: row t of un_ss_table caches all of them (kUnSsSteps = diffusion_steps rows of un_ss_total floats, ~54 MB, plus
    // one scratch row for steps beyond the schedule), filled the first time a step is seen — 24 small launches, ~1 ms, per step
    // instead of per network evaluation.
    size_t un_ss_total = 0;
    float* un_ss_table = nullptr;
    const float* un_ss_cur = nullptr;
    std::vector<char> un_ss_have;
    float *un_te0w = nullptr, *un_te0b = nullptr, *un_te2w = nullptr, *un_te2b = nullptr, *un_outgw = nullptr, *un_outgb = nullptr;
    float *un_outw = nullptr, *un_outb = nullptr, *un_temb = nullptr, *un_emb1 = nullptr, *un_emb = nullptr, *un_semb = nullptr;
    float* un_buf[8] = {nullptr};          // work maps: 3 rotating block outputs, T1, T2, skip, qkv / cat, attention
    float* un_eps = nullptr;
    // 16-bit tier of the UNet (gemm_h16: f16 operands, fp32 accumulate; GroupNorm / softmax / residual sums stay fp32): f16 twins of
    // the block outputs (the maps a GEMM reads without a GroupNorm in between), f16-only GroupNorm / upsample / attention outputs
    bool un_h16 = false;
    bool un_x3 = false;                    // exact-vote engines: the UNet's middle tier (fp32 pipeline on split-f16 operands, gemm_x3_kernel) is resident
    h16_t* un_buf16[3] = {nullptr};
    std::vector<h16_t*> un_hs16;
    h16_t *un_t1h = nullptr, *un_t2h = nullptr, *un_uph = nullptr, *un_atth = nullptr, *un_qkvh = nullptr;
    // GroupNorm statistics of the f16 maps, written by the GEMM that produces the map (GemmH16Args::stats): one slab per map buffer
    float* un_st_buf[3] = {nullptr};
    float* un_st_t2 = nullptr;
    std::vector<float*> un_st_hs;
    float tau_spec = 0.f;                  // recheck bound of the spec-domain vote loop's 16-bit tier (dmad_set_spec_recheck_margin)
    float tau_spec2 = 0.f;                 // ... and of its split-f16 tier (dmad_set_spec_recheck_margin2; < 0: no middle tier)
    int64_t st_spec_samples = 0, st_spec_rechecked = 0, st_spec_rechecked2 = 0;

    template <typename T>
    int alloc(T** p, size_t n, bool zero = false) {
        void* d = nullptr;
        hipError_t e = hipMalloc(&d, n * sizeof(T));
        if (e != hipSuccess) return fail(DMAD_ERR_HIP, "hipMalloc(%zu bytes) failed: %s", n * sizeof(T), hipGetErrorString(e));
        if (zero) {
            e = hipMemset(d, 0, n * sizeof(T));
            if (e != hipSuccess) return fail(DMAD_ERR_HIP, "hipMemset failed: %s", hipGetErrorString(e));
        }
        allocs.push_back(d);
        bytes += (int64_t)(n * sizeof(T));
        *p = (T*)d;
        return 0;
    }
    template <typename T>
    int upload(T** p, const std::vector<T>& h) {
        CHK(alloc(p, h.size()));
        HIPCHK(hipMemcpy(*p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
        return 0;
    }
    int upload_bf(h16_t** p, const std::vector<uint16_t>& h) {
        CHK(alloc(p, h.size()));
        HIPCHK(hipMemcpy(*p, h.data(), h.size() * 2, hipMemcpyHostToDevice));
        return 0;
    }
    const HostW* get(const std::string& name, std::initializer_list<int64_t> shape) {
        auto it = hw.find(name);
        if (it == hw.end()) {
            fail(DMAD_ERR_STATE, "weight '%s' was not loaded", name.c_str());
            return nullptr;
        }
        int64_t n = 1;
        for (auto s : shape) n *= s;
        if ((int64_t)it->second.v.size() != n) {
            fail(DMAD_ERR_INVALID, "weight '%s' has %zu elements, expected %lld", name.c_str(), it->second.v.size(), (long long)n);
            return nullptr;
        }
        return &it->second;
    }
};

namespace {

// [ksteps][rows][32] bf16 LDS image of W[row][K] (row-major, K = ksteps*32), 64-B rows, swz64 chunks
// k-step ks of W lands in stage ks * smul + sadd of the image (GEMM1 interleaves the three taps' k-steps)
void pack_rows(uint16_t (*cvt)(float), const float* W, int rows, int K, long ldw, const int* row_map, std::vector<uint16_t>& out,
               size_t base, int smul = 1, int sadd = 0) {
    const int ksteps = K / 32;
    for (int ks = 0; ks < ksteps; ++ks)
        for (int R = 0; R < rows; ++R) {
            const float* src = W + (long)(row_map ? row_map[R] : R) * ldw + ks * 32;
            for (int slot = 0; slot < 4; ++slot) {
                const int c = slot ^ swz64(R);
                for (int j = 0; j < 8; ++j) out[base + ((size_t)((ks * smul + sadd) * rows + R) * 32) + slot * 8 + j] = cvt(src[c * 8 + j]);
            }
        }
}

int finalize_wavenet(dmad_engine* e) {
    const int NL = e->NL;
    const HostW* w;
#define GETW(var, name, ...)                     \
    w = e->get(name, {__VA_ARGS__});             \
    if (!w) return DMAD_ERR_STATE;               \
    const std::vector<float>& var = w->v;
    GETW(init_w, "init.w", 256) GETW(init_b, "init.b", 256)
    GETW(fc1w, "fc_t1.w", 512, 128) GETW(fc1b, "fc_t1.b", 512)
    GETW(fc2w, "fc_t2.w", 512, 512) GETW(fc2b, "fc_t2.b", 512)
    GETW(f0w, "f0.w", 256, 256) GETW(f0b, "f0.b", 256)
    GETW(f2w, "f2.w", 256) GETW(f2b, "f2.b", 1)
    CHK(e->upload(&e->init_w, init_w)); CHK(e->upload(&e->init_b, init_b));
    CHK(e->upload(&e->fc1w, fc1w)); CHK(e->upload(&e->fc1b, fc1b));
    CHK(e->upload(&e->fc2w, fc2w)); CHK(e->upload(&e->fc2b, fc2b));
    CHK(e->upload(&e->bf0, f0b)); CHK(e->upload(&e->wz, f2w));
    e->bz = f2b[0];
    std::vector<float> fctw((size_t)NL * 256 * 512), fctb((size_t)NL * 256);
    for (int n = 0; n < NL; ++n) {
        char nm[64];
        snprintf(nm, sizeof nm, "fc_t.%d.w", n);
        GETW(a, nm, 256, 512)
        memcpy(&fctw[(size_t)n * 256 * 512], a.data(), a.size() * 4);
        snprintf(nm, sizeof nm, "fc_t.%d.b", n);
        GETW(bb, nm, 256)
        memcpy(&fctb[(size_t)n * 256], bb.data(), 1024);
    }
    CHK(e->upload(&e->fctw, fctw)); CHK(e->upload(&e->fctb, fctb));
    CHK(e->alloc(&e->emb_table, (size_t)NL * 256)); CHK(e->alloc(&e->emb2, 512)); CHK(e->alloc(&e->epi_c, (size_t)NL * 256, true));

    if (e->bf16) {
        uint16_t (*cvt)(float) = e->f16 ? f2h_census : f2bf;
        g_h_sq = g_h_sq_sub = g_h_worst = 0.0; g_h_bad_tensors = g_h_ovf = g_h_tensors = 0;
#ifdef DMAD_DEV_WEIGHT_MASK
        int mask_bits = 0, mask_which = 3;              // development build only, see f2h_census_masked
        if (const char* mb = getenv("DMAD_WEIGHT_MASK_BITS")) mask_bits = atoi(mb);
        if (const char* mw = getenv("DMAD_WEIGHT_MASK_WHICH")) mask_which = atoi(mw);
        if (mask_bits < 0 || mask_bits > 9 || !e->f16) mask_bits = 0;
        if (mask_bits) e->warn = "DEVELOPMENT BUILD: the f16 WaveNet weight images are rounded to fewer mantissa bits (DMAD_WEIGHT_MASK_BITS); the recheck bounds do not hold";
        auto cvt_for = [&](int which) -> uint16_t (*)(float) {
            g_mask_bits = mask_bits;
            return (mask_bits && (mask_which >> which & 1)) ? f2h_census_masked : cvt;
        };
#else
        auto cvt_for = [&](int) -> uint16_t (*)(float) { return cvt; };
#endif
        int rmap[512];
        // tile row R = wm*128 + half*64 + mt*16 + i  <->  gate row half*256 + (mt*64 + wm*16 + i): channel ownership is
        // interleaved over the M-waves so that GEMM2 can start on channels [64 mt, 64 mt + 64) as soon as tiles mt are gated
        for (int R = 0; R < 512; ++R) rmap[R] = ((R % 128) / 64) * 256 + ((R % 64) / 16) * 64 + (R / 128) * 16 + (R % 16);
        std::vector<uint16_t> w1p((size_t)NL * 24 * 512 * 32), w2p((size_t)NL * 8 * 256 * 32), wsp((size_t)NL * 8 * 256 * 32),
            wf0p((size_t)8 * 256 * 32);
        std::vector<float> b1p((size_t)NL * 512), b2((size_t)NL * 256), bsum(256, 0.f), tapw((size_t)512 * 256);
        for (int n = 0; n < NL; ++n) {
            char nm[64];
            snprintf(nm, sizeof nm, "dil.%d.w", n); GETW(dw, nm, 512, 256, 3)
            snprintf(nm, sizeof nm, "dil.%d.b", n); GETW(db, nm, 512)
            snprintf(nm, sizeof nm, "res.%d.w", n); GETW(rw, nm, 256, 256)
            snprintf(nm, sizeof nm, "res.%d.b", n); GETW(rb, nm, 256)
            snprintf(nm, sizeof nm, "skip.%d.w", n); GETW(sw, nm, 256, 256)
            snprintf(nm, sizeof nm, "skip.%d.b", n); GETW(sb, nm, 256)
            for (int tap = 0; tap < 3; ++tap) {
                for (int oc = 0; oc < 512; ++oc)
                    for (int ci = 0; ci < 256; ++ci)     // rows pre-scaled to exp2 arguments: tanh half by -2*log2(e), sigmoid half by -log2(e)
                        tapw[(size_t)oc * 256 + ci] = dw[((size_t)oc * 256 + ci) * 3 + tap] * (oc < 256 ? -2.8853900817779268f : -1.4426950408889634f);
                pack_rows(cvt_for(0), tapw.data(), 512, 256, 256, rmap, w1p, (size_t)n * 24 * 512 * 32, 3, tap);   // stage = 3 * kchunk + tap
            }
            census_close_tensor();
            for (int R = 0; R < 512; ++R) b1p[(size_t)n * 512 + R] = db[rmap[R]];
            std::vector<float> rws(rw.size());                      // res conv pre-scaled by sqrt(1/2): h' = h*sqrt(1/2) + (W_res' g + c)
            for (size_t i = 0; i < rw.size(); ++i) rws[i] = rw[i] * 0.70710678118654752440f;
            pack_rows(cvt_for(1), rws.data(), 256, 256, 256, nullptr, w2p, (size_t)n * 8 * 256 * 32);
            census_close_tensor();
            pack_rows(cvt_for(2), sw.data(), 256, 256, 256, nullptr, wsp, (size_t)n * 8 * 256 * 32);
            census_close_tensor();
            for (int c = 0; c < 256; ++c) { b2[(size_t)n * 256 + c] = rb[c]; bsum[c] += sb[c]; }
        }
        pack_rows(cvt_for(3), f0w.data(), 256, 256, 256, nullptr, wf0p, 0);
        census_close_tensor();
        CHK(e->upload_bf(&e->w1p, w1p)); CHK(e->upload_bf(&e->w2p, w2p)); CHK(e->upload_bf(&e->wsp, wsp));
        CHK(e->upload_bf(&e->wf0p, wf0p));
        CHK(e->upload(&e->b1p, b1p)); CHK(e->upload(&e->b2, b2)); CHK(e->upload(&e->bskip_sum, bsum));
        if (e->f16 && (g_h_bad_tensors || g_h_ovf)) {
            char buf[448];
            snprintf(buf, sizeof buf, "WaveNet weights on the f16 MFMA path: in %ld of %ld folded weight tensors f16 subnormals (|w| < 6.1e-5: fewer "
                     "than 11 significant bits) carry more than 1e-6 of the squared norm (worst: %.3g), and %ld values overflow to inf "
                     "(|w| > 65504); the 16-bit tier's error bound was not measured for such weights: calibrate the recheck margins on "
                     "these weights or use half_type = bf16", g_h_bad_tensors, g_h_tensors, g_h_worst, g_h_ovf);
            e->warn = buf;
        }
    }
    if (e->f32) {
        std::vector<float> wdil((size_t)NL * 3 * 512 * 256), bdil((size_t)NL * 512), wrs((size_t)NL * 512 * 256), brs((size_t)NL * 512);
        for (int n = 0; n < NL; ++n) {
            char nm[64];
            snprintf(nm, sizeof nm, "dil.%d.w", n); GETW(dw, nm, 512, 256, 3)
            snprintf(nm, sizeof nm, "dil.%d.b", n); GETW(db, nm, 512)
            snprintf(nm, sizeof nm, "res.%d.w", n); GETW(rw, nm, 256, 256)
            snprintf(nm, sizeof nm, "res.%d.b", n); GETW(rb, nm, 256)
            snprintf(nm, sizeof nm, "skip.%d.w", n); GETW(sw, nm, 256, 256)
            snprintf(nm, sizeof nm, "skip.%d.b", n); GETW(sb, nm, 256)
            // gate-fused epilogue (gemm_f32.h, epi 1): image row R of block bm holds H row (i >= 2 ? 256 : 0) + bm*64 + wm*32 + (i&1)*16 + r
            for (int tap = 0; tap < 3; ++tap)
                for (int R = 0; R < 512; ++R) {
                    const int bm = R / 128, wmr = (R % 128) / 64, i = (R % 64) / 16, rr = R % 16;
                    const int oc = (i >= 2 ? 256 : 0) + bm * 64 + wmr * 32 + (i & 1) * 16 + rr;
                    for (int ci = 0; ci < 256; ++ci)
                        wdil[(((size_t)n * 3 + tap) * 512 + R) * 256 + ci] = dw[((size_t)oc * 256 + ci) * 3 + tap];
                    if (tap == 0) bdil[(size_t)n * 512 + R] = db[oc];
                }
            memcpy(&wrs[(size_t)n * 512 * 256], rw.data(), 256 * 256 * 4);
            memcpy(&wrs[(size_t)n * 512 * 256 + 256 * 256], sw.data(), 256 * 256 * 4);
            memcpy(&brs[(size_t)n * 512], rb.data(), 1024);
            memcpy(&brs[(size_t)n * 512 + 256], sb.data(), 1024);
        }
        CHK(e->upload(&e->wdil, wdil)); CHK(e->upload(&e->bdil, bdil)); CHK(e->upload(&e->wrs, wrs)); CHK(e->upload(&e->brs, brs));
        CHK(e->upload(&e->wf0, f0w));
        // the NL skip convs run as ONE K = NL * 256 GEMM over the stored gate outputs of all layers (as the 16-bit path does):
        // no fp32 read-modify-write of the skip sum per layer
        std::vector<float> ws((size_t)NL * 256 * 256), bs(256, 0.f);
        for (int n = 0; n < NL; ++n) {
            memcpy(&ws[(size_t)n * 256 * 256], &wrs[(size_t)n * 512 * 256 + 256 * 256], 256 * 256 * 4);
            for (int c = 0; c < 256; ++c) bs[c] += brs[(size_t)n * 512 + 256 + c];
        }
        CHK(e->upload(&e->wskip32, ws));
        CHK(e->upload(&e->bskip32, bs));
        CHK(e->alloc(&e->gstore32, (size_t)NL * e->maxB32 * e->L * 256));      // [NL][maxB32 * L][256], shared by the fp32 and split-f16 tiers
        if (e->bf16) {                      // exact-vote engines: the middle (split-f16, three-MFMA) tier reads these
            std::vector<float> t(wdil.size());
            split_rows(wdil.data(), wdil.size(), t.data()); CHK(e->upload(&e->wdil_x3, t));
            t.resize(wrs.size()); split_rows(wrs.data(), wrs.size(), t.data()); CHK(e->upload(&e->wrs_x3, t));
            t.resize(f0w.size()); split_rows(f0w.data(), f0w.size(), t.data()); CHK(e->upload(&e->wf0_x3, t));
            t.resize(ws.size()); split_rows(ws.data(), ws.size(), t.data()); CHK(e->upload(&e->wskip_x3, t));
        }
    }
    return 0;
}

int init_mel_constants(dmad_engine* e) {
    // mel constants, float64 on the host then rounded once (torchaudio MelSpectrogram semantics)
    {
        std::vector<float> A((size_t)kDftM * 2048);
        std::vector<double> win(2048);
        for (int n = 0; n < 2048; ++n) win[n] = 0.5 - 0.5 * cos(2.0 * M_PI * n / 2048.0);
        for (int f = 0; f < 1025; ++f)
            for (int n = 0; n < 2048; ++n) {
                const long ph = ((long)f * n) % 2048;                 // exact argument reduction
                const double ang = 2.0 * M_PI * (double)ph / 2048.0;
                A[(size_t)f * 2048 + n] = (float)(win[n] * cos(ang));
                A[(size_t)(1025 + f) * 2048 + n] = (float)(-win[n] * sin(ang));
            }
        CHK(e->upload(&e->dftA, A));
        // slaney mel filterbank, [32][kMelLd]
        auto hz2mel = [](double f) { return f >= 1000.0 ? 15.0 + log(f / 1000.0) / (log(6.4) / 27.0) : f / (200.0 / 3); };
        auto mel2hz = [](double m) { return m >= 15.0 ? 1000.0 * exp((log(6.4) / 27.0) * (m - 15.0)) : (200.0 / 3) * m; };
        double fpts[34];
        const double m0 = hz2mel(0.0), m1 = hz2mel(8000.0);
        for (int i = 0; i < 34; ++i) fpts[i] = mel2hz(m0 + (m1 - m0) * i / 33.0);
        std::vector<float> fb((size_t)32 * kMelLd, 0.f);
        for (int m = 0; m < 32; ++m) {
            const double enorm = 2.0 / (fpts[m + 2] - fpts[m]);
            for (int f = 0; f < 1025; ++f) {
                const double fr = 8000.0 * f / 1024.0;
                const double down = (fr - fpts[m]) / (fpts[m + 1] - fpts[m]);
                const double up = (fpts[m + 2] - fr) / (fpts[m + 2] - fpts[m + 1]);
                const double v = fmax(0.0, fmin(down, up));
                fb[(size_t)m * kMelLd + f] = (float)(v * enorm);
            }
        }
        CHK(e->upload(&e->fbA, fb));
    }
    return 0;
}

int finalize_classifier(dmad_engine* e) {
    const HostW* w;
    // VGG19_bn
    int cin = 1, li = 0;
    for (int i = 0; i < kVggCfgLen; ++i) {
        const int v = kVggCfg[i];
        if (v < 0) continue;
        char nm[64];
        snprintf(nm, sizeof nm, "vgg.conv%d.w", li);
        w = e->get(nm, {v, cin, 3, 3});
        if (!w) return DMAD_ERR_STATE;
        const std::vector<float>& cw = w->v;
        if (li == 0) {
            CHK(e->upload(&e->vconv1w, cw));
        } else {
            std::vector<float> A((size_t)9 * v * cin);
            for (int co = 0; co < v; ++co)
                for (int ci = 0; ci < cin; ++ci)
                    for (int t = 0; t < 9; ++t) A[((size_t)t * v + co) * cin + ci] = cw[((size_t)co * cin + ci) * 9 + t];
            CHK(e->upload(&e->vconvw[li], A));
        }
        snprintf(nm, sizeof nm, "vgg.conv%d.scale", li);
        w = e->get(nm, {v}); if (!w) return DMAD_ERR_STATE;
        CHK(e->upload(&e->vscale[li], w->v));
        snprintf(nm, sizeof nm, "vgg.conv%d.shift", li);
        w = e->get(nm, {v}); if (!w) return DMAD_ERR_STATE;
        CHK(e->upload(&e->vshift[li], w->v));
        cin = v;
        ++li;
    }
    const int fin[3] = {512, 4096, 4096}, fout[3] = {4096, 4096, e->cfg.num_classes};
    for (int j = 0; j < 3; ++j) {
        char nm[64];
        snprintf(nm, sizeof nm, "vgg.fc%d.w", j);
        w = e->get(nm, {fout[j], fin[j]}); if (!w) return DMAD_ERR_STATE;
        CHK(e->upload(&e->vfcw[j], w->v));
        snprintf(nm, sizeof nm, "vgg.fc%d.b", j);
        w = e->get(nm, {fout[j]}); if (!w) return DMAD_ERR_STATE;
        CHK(e->upload(&e->vfcb[j], w->v));
    }
    return 0;
}

GemmF32Args plain_gemm(const float* A, const float* X, float* C, const float* scale, const float* shift, int M, int K, long N,
                       int ldc, long ldx, int relu);
int upload_split(dmad_engine* e, const std::vector<float>& A, float** wx);

// ResNeXt29 8x64d: names rx.conv1.*, rx.b<i>.{reduce,conv,expand,short}.{w,scale,shift} (i = 3 * stage + bottleneck),
// rx.fc.{w,b}.  GEMM images: 1x1 convs [M][K]; the grouped 3x3 conv per group [tap][M/8][K/8] (models/resnext.py:23-62).
int finalize_resnext(dmad_engine* e) {
    const HostW* w;
    // A: the fp32 GEMM image; rows_of(i) = the output channel whose BN scale multiplies element i of the f16 image Ah (null: Ah = A)
    auto up3 = [&](const std::string& base, dmad_engine::RxConv& c, const std::vector<float>& A, int M, const std::vector<float>* Ah = nullptr,
                   const std::vector<int>* ch_of = nullptr, long row_len = 0) -> int {
        CHK(e->upload(&c.w, A));
        w = e->get(base + ".scale", {M}); if (!w) return DMAD_ERR_STATE;
        const std::vector<float> scale = w->v;
        CHK(e->upload(&c.scale, scale));
        w = e->get(base + ".shift", {M}); if (!w) return DMAD_ERR_STATE;
        CHK(e->upload(&c.shift, w->v));
        if (e->rx_h16 && row_len > 0) {     // f16 image, BN scale folded into the rows (the 16-bit kernel's epilogue only adds the shift)
            const std::vector<float>& S = Ah ? *Ah : A;
            std::vector<uint16_t> Hh(S.size());
            for (size_t i = 0; i < S.size(); ++i) {
                const long row = (long)(i / (size_t)row_len);
                const int ch = ch_of ? (*ch_of)[row] : (int)(row % M);
                Hh[i] = f2h(ch >= 0 ? S[i] * scale[ch] : 0.f);
            }
            CHK(e->upload_bf(&c.wh, Hh));
        }
        if (e->rx_x3 && row_len > 0) CHK(upload_split(e, Ah ? *Ah : A, &c.wx));      // the same image (grouped conv: the paired block-diagonal one), BN scale / shift stay in the epilogue
        return 0;
    };
    w = e->get("rx.conv1.w", {64, 1, 3, 3}); if (!w) return DMAD_ERR_STATE;
    CHK(up3("rx.conv1", e->rxconv1, w->v, 64));
    const int stages[4] = {64, 256, 512, 1024};
    for (int i = 0; i < 9; ++i) {
        dmad_engine::RxBlock& b = e->rx[i];
        const int st = i / 3, k = i % 3;
        b.cin = k == 0 ? stages[st] : stages[st + 1];
        b.cout = stages[st + 1];
        b.D = 8 * (64 * b.cout / 256);
        b.stride = (k == 0 && st > 0) ? 2 : 1;
        b.has_short = b.cin != b.cout;
        const std::string base = "rx.b" + std::to_string(i);
        w = e->get(base + ".reduce.w", {b.D, b.cin}); if (!w) return DMAD_ERR_STATE;
        CHK(up3(base + ".reduce", b.reduce, w->v, b.D, nullptr, nullptr, b.cin));
        const int G = b.D / 8;
        w = e->get(base + ".conv.w", {b.D, G, 3, 3}); if (!w) return DMAD_ERR_STATE;
        std::vector<float> A((size_t)8 * 9 * G * G);
        for (int g = 0; g < 8; ++g)
            for (int m = 0; m < G; ++m)
                for (int kk = 0; kk < G; ++kk)
                    for (int t = 0; t < 9; ++t)
                        A[(((size_t)g * 9 + t) * G + m) * G + kk] = w->v[(((size_t)g * G + m) * G + kk) * 9 + t];
        {   // f16 image of the grouped conv: [group][tap][Mg][Kg].  gemm_h16 needs Mg % 128 == 0: the 64-channel groups of stage 1
            // are paired into 128 x 128 block-diagonal groups (the off-diagonal blocks are zeros: twice the MFMAs on 17 % of the network)
            const int pair = G < 128 ? 2 : 1, Gp = G * pair, ngp = 8 / pair;
            std::vector<float> Ah((size_t)ngp * 9 * Gp * Gp, 0.f);
            std::vector<int> ch_of((size_t)ngp * 9 * Gp);
            for (int gp = 0; gp < ngp; ++gp)
                for (int t = 0; t < 9; ++t)
                    for (int m = 0; m < Gp; ++m) {
                        const int g = gp * pair + m / G, mm = m % G;
                        ch_of[((size_t)gp * 9 + t) * Gp + m] = g * G + mm;
                        for (int kk = 0; kk < G; ++kk)
                            Ah[(((size_t)gp * 9 + t) * Gp + m) * Gp + (m / G) * G + kk] = w->v[(((size_t)g * G + mm) * G + kk) * 9 + t];
                    }
            CHK(up3(base + ".conv", b.conv, A, b.D, &Ah, &ch_of, Gp));
        }
        w = e->get(base + ".expand.w", {b.cout, b.D}); if (!w) return DMAD_ERR_STATE;
        CHK(up3(base + ".expand", b.expand, w->v, b.cout, nullptr, nullptr, b.D));
        if (b.has_short) {
            w = e->get(base + ".short.w", {b.cout, b.cin}); if (!w) return DMAD_ERR_STATE;
            CHK(up3(base + ".short", b.shortc, w->v, b.cout, nullptr, nullptr, b.cin));
        }
    }
    w = e->get("rx.fc.w", {e->cfg.num_classes, 1024}); if (!w) return DMAD_ERR_STATE;
    CHK(e->upload(&e->rxfcw, w->v));
    w = e->get("rx.fc.b", {e->cfg.num_classes}); if (!w) return DMAD_ERR_STATE;
    CHK(e->upload(&e->rxfcb, w->v));
    const size_t B = (size_t)e->maxB;
    CHK(e->alloc(&e->rxX, B * 1024 * 256));
    CHK(e->alloc(&e->rxY, B * 1024 * 256));
    CHK(e->alloc(&e->rxS, B * 1024 * 256));
    CHK(e->alloc(&e->rxT1, B * 1024 * 1024));     // stage 2's first reduce: 32x32 pixels x D = 1024
    CHK(e->alloc(&e->rxT2, B * 1024 * 512));
    if (e->rx_h16) {
        CHK(e->alloc(&e->rxX16, B * 1024 * 256)); CHK(e->alloc(&e->rxY16, B * 1024 * 256)); CHK(e->alloc(&e->rxS16, B * 1024 * 256));
        CHK(e->alloc(&e->rxT1h, B * 1024 * 1024)); CHK(e->alloc(&e->rxT2h, B * 1024 * 512));
        if (int r = gemm_h16_configure()) return fail(DMAD_ERR_HIP, "hipFuncSetAttribute(max dynamic LDS, f16 conv GEMM) failed: %d", r);
    }
    if (e->rx_x3) if (int r = gemm_x3_configure()) return fail(DMAD_ERR_HIP, "hipFuncSetAttribute(max dynamic LDS, split-f16 tier) failed: %d", r);
    return 0;
}

// The same network on its SPLIT-F16 tier (exact-vote engines): the fp32 pipeline's structure with every conv on split-f16 operands (three f16
// MFMAs per product, ~22 significant bits: gemm_x3_kernel's NHWC form, grouped for the 3x3), the eval-mode BatchNorm scale / shift, the
// shortcut add and the ReLU in the fp32 epilogue; a map that only feeds GEMMs (and, as a block output, the next shortcut add) is written
// in the split format directly.  Stage 1's 64-channel groups are paired into 128 x 128 block-diagonal groups (the kernel's tiles need
// 128 rows), like on the 16-bit tier.  Average pool and head in fp32.
int classify_resnext_x3(dmad_engine* e, const float* spec, int B, float* logits, hipStream_t s) {
    float *X = e->rxX, *Y = e->rxY;
    launch_vgg_conv1(spec, e->rxconv1.w, e->rxconv1.scale, e->rxconv1.shift, X, B, s);      // 1 -> 64, 3x3, BN, ReLU (direct kernel, fp32)
    launch_scale(X, 1.f, X, (long)B * 1024 * 64, s, true);                                   // ... as a split-format operand, in place
    int H = 32;
    for (int i = 0; i < 9; ++i) {
        const dmad_engine::RxBlock& b = e->rx[i];
        const int Ho = (H - 1) / b.stride + 1;
        const long Nin = (long)B * H * H, Nout = (long)B * Ho * Ho;
        auto mk = [&](const dmad_engine::RxConv& c, const float* in, float* out, int M, int K, int taps, long N, int Hin, int ldx, int ldc, int stride, int relu,
                      int out_split) {
            GemmF32Args g{};
            g.A = c.wx; g.X = in; g.C = out; g.scale = c.scale; g.shift = c.shift; g.M = M; g.K = K; g.taps = taps; g.ldc = ldc; g.relu = relu; g.N = N;
            g.mode = 2; g.H = Hin; g.W = Hin; g.Cin = K; g.ldx = ldx; g.stride = stride; g.x3 = 1; g.out_split = out_split;
            return g;
        };
        launch_gemm_f32(mk(b.reduce, X, e->rxT1, b.D, b.cin, 1, Nin, H, b.cin, b.D, 1, 1, 1), s);                 // conv_reduce + bn + ReLU
        const int G = b.D / 8, pair = G < 128 ? 2 : 1;
        GemmF32Args c = mk(b.conv, e->rxT1, e->rxT2, G * pair, G * pair, 9, Nout, H, b.D, b.D, b.stride, 1, 1);    // grouped 3x3 (stride) + bn + ReLU
        c.groups = 8 / pair;
        launch_gemm_f32(c, s);
        GemmF32Args x = mk(b.expand, e->rxT2, Y, b.cout, b.D, 1, Nout, Ho, b.D, b.cout, 1, 1, i == 8 ? 0 : 1);    // conv_expand + bn + shortcut, ReLU
        if (b.has_short) {
            launch_gemm_f32(mk(b.shortc, X, e->rxS, b.cout, b.cin, 1, Nout, H, b.cin, b.cout, b.stride, 0, 0), s);  // shortcut conv + bn: fp32
            x.res = e->rxS;
        } else {
            x.res = X; x.res_split = 1;                                                                              // the block input exists in the split format only
        }
        launch_gemm_f32(x, s);
        float* t = X; X = Y; Y = t;
        H = Ho;
    }
    launch_avgpool_nhwc(X, e->rxT2, B, H * H, 1024, s);
    launch_gemm_f32(plain_gemm(e->rxfcw, e->rxT2, logits, nullptr, e->rxfcb, e->cfg.num_classes, 1024, B, e->cfg.num_classes, 1024, 0), s,
                    e->slab, e->slab_floats, (long)e->maxB);
    LASTCHK();
    return 0;
}

// The same network on its 16-bit tier: f16 maps, every conv through gemm_h16 (BN scale folded into the f16 weights, shift / shortcut /
// ReLU in the fp32 epilogue); the last bottleneck writes fp32 for the average pool and the fp32 classifier head.
int classify_resnext_h16(dmad_engine* e, const float* spec, int B, float* logits, hipStream_t s) {
    h16_t *X = e->rxX16, *Y = e->rxY16;
    launch_vgg_conv1(spec, e->rxconv1.w, e->rxconv1.scale, e->rxconv1.shift, nullptr, B, s, X);
    int H = 32;
    for (int i = 0; i < 9; ++i) {
        const dmad_engine::RxBlock& b = e->rx[i];
        const int Ho = (H - 1) / b.stride + 1;
        const long Nin = (long)B * H * H, Nout = (long)B * Ho * Ho;
        auto mk = [&](const dmad_engine::RxConv& c, const h16_t* in, h16_t* out16, int M, int K, int taps, long N, int Hin, int ldx, int ldc, int stride) {
            GemmH16Args g{};
            g.A = c.wh; g.X = in; g.C16 = out16; g.shift = c.shift; g.M = M; g.K = K; g.taps = taps; g.ldc = ldc; g.N = N; g.H = Hin; g.W = Hin;
            g.ldx = ldx; g.stride = stride; g.relu = 1;
            return g;
        };
        launch_gemm_h16(mk(b.reduce, X, e->rxT1h, b.D, b.cin, 1, Nin, H, b.cin, b.D, 1), s);                      // conv_reduce + bn + ReLU
        const int G = b.D / 8, pair = G < 128 ? 2 : 1;
        GemmH16Args c = mk(b.conv, e->rxT1h, e->rxT2h, G * pair, G * pair, 9, Nout, H, b.D, b.D, b.stride);         // grouped 3x3 (stride) + bn + ReLU
        c.groups = 8 / pair;
        launch_gemm_h16(c, s);
        const h16_t* res = X;
        if (b.has_short) {
            GemmH16Args h = mk(b.shortc, X, e->rxS16, b.cout, b.cin, 1, Nout, H, b.cin, b.cout, b.stride);         // shortcut conv + bn (no ReLU)
            h.relu = 0;
            launch_gemm_h16(h, s);
            res = e->rxS16;
        }
        GemmH16Args x = mk(b.expand, e->rxT2h, i == 8 ? nullptr : Y, b.cout, b.D, 1, Nout, Ho, b.D, b.cout, 1);   // conv_expand + bn + shortcut, ReLU
        x.res16 = res;
        if (i == 8) x.C = e->rxY;
        launch_gemm_h16(x, s);
        h16_t* t = X; X = Y; Y = t;
        H = Ho;
    }
    launch_avgpool_nhwc(e->rxY, e->rxT2, B, H * H, 1024, s);
    launch_gemm_f32(plain_gemm(e->rxfcw, e->rxT2, logits, nullptr, e->rxfcb, e->cfg.num_classes, 1024, B, e->cfg.num_classes, 1024, 0), s,
                    e->slab, e->slab_floats, (long)e->maxB);
    LASTCHK();
    return 0;
}

// CifarResNeXt.forward (models/resnext.py:133-142) on NHWC fp32 maps
int classify_resnext(dmad_engine* e, const float* spec, int B, float* logits, hipStream_t s) {
    float *X = e->rxX, *Y = e->rxY;
    launch_vgg_conv1(spec, e->rxconv1.w, e->rxconv1.scale, e->rxconv1.shift, X, B, s);      // 1 -> 64, 3x3, BN, ReLU
    int H = 32;
    for (int i = 0; i < 9; ++i) {
        const dmad_engine::RxBlock& b = e->rx[i];
        const int Ho = (H - 1) / b.stride + 1;
        const long Nin = (long)B * H * H, Nout = (long)B * Ho * Ho, nref_in = (long)e->maxB * H * H, nref_out = (long)e->maxB * Ho * Ho;
        // conv_reduce + bn_reduce + ReLU (1x1)
        GemmF32Args g = plain_gemm(b.reduce.w, X, e->rxT1, b.reduce.scale, b.reduce.shift, b.D, b.cin, Nin, b.D, b.cin, 1);
        launch_gemm_f32(g, s, e->slab, e->slab_floats, nref_in);
        // conv_conv (3x3, 8 groups, stride) + bn + ReLU
        GemmF32Args c{};
        c.A = b.conv.w; c.X = e->rxT1; c.C = e->rxT2; c.scale = b.conv.scale; c.shift = b.conv.shift;
        c.M = b.D / 8; c.K = b.D / 8; c.taps = 9; c.ldc = b.D; c.relu = 1; c.N = Nout; c.mode = 2;
        c.H = H; c.W = H; c.Cin = b.D / 8; c.ldx = b.D; c.stride = b.stride; c.groups = 8;
        launch_gemm_f32(c, s);
        // shortcut: identity, or 1x1 conv (stride) + BN
        const float* res = X;
        if (b.has_short) {
            GemmF32Args h{};
            h.A = b.shortc.w; h.X = X; h.C = e->rxS; h.scale = b.shortc.scale; h.shift = b.shortc.shift;
            h.M = b.cout; h.K = b.cin; h.taps = 1; h.ldc = b.cout; h.relu = 0; h.N = Nout; h.mode = 2;
            h.H = H; h.W = H; h.Cin = b.cin; h.ldx = b.cin; h.stride = b.stride;
            launch_gemm_f32(h, s, e->slab, e->slab_floats, nref_out);
            res = e->rxS;
        }
        // conv_expand + bn_expand, + shortcut, ReLU
        GemmF32Args x = plain_gemm(b.expand.w, e->rxT2, Y, b.expand.scale, b.expand.shift, b.cout, b.D, Nout, b.cout, b.D, 1);
        x.res = res;
        launch_gemm_f32(x, s, e->slab, e->slab_floats, nref_out);
        float* t = X; X = Y; Y = t;
        H = Ho;
    }
    launch_avgpool_nhwc(X, e->rxT2, B, H * H, 1024, s);
    launch_gemm_f32(plain_gemm(e->rxfcw, e->rxT2, logits, nullptr, e->rxfcb, e->cfg.num_classes, 1024, B, e->cfg.num_classes, 1024, 0), s,
                    e->slab, e->slab_floats, (long)e->maxB);
    LASTCHK();
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------
// Improved-Diffusion UNet.  Configuration of the reference's wrapper (improved_diffusion_ddpm.py:
// 64-93 + script_util.py:11-34,100-131): 1 -> 128 channels, 3 ResBlocks per level, channel_mult (1,2,2,2), attention at
// 16x16 and 8x8 (+ the middle block), 4 heads, scale-shift norm, epsilon output.  Weight names = "un." + the reference's
// state-dict names.  Every conv / linear is a gemm_f32 launch over NHWC maps.
// ---------------------------------------------------------------------------------------------------------------
constexpr int kUnMC = 128, kUnTE = 512, kUnHeads = 4, kUnRes = 3;
constexpr int kUnSsSteps = 1000;         // cached steps (create_improved_diffusion: diffusion_steps = 1000)
const int kUnMult[4] = {1, 2, 2, 2};
inline bool un_attn_at(int ds) { return ds == 2 || ds == 4; }

int upload_h16(dmad_engine* e, const std::vector<float>& A, h16_t** wh) {
    std::vector<uint16_t> H(A.size());
    for (size_t i = 0; i < A.size(); ++i) H[i] = f2h(A[i]);
    return e->upload_bf(wh, H);
}

int upload_split(dmad_engine* e, const std::vector<float>& A, float** wx) {
    std::vector<float> t(A.size());
    split_rows(A.data(), A.size(), t.data());
    return e->upload(wx, t);
}

int un_conv3(dmad_engine* e, const std::string& name, int cout, int cin, float** w, float** b, h16_t** wh = nullptr, float** wx = nullptr) {
    const HostW* h = e->get(name + ".weight", {cout, cin, 3, 3}); if (!h) return DMAD_ERR_STATE;
    std::vector<float> A((size_t)9 * cout * cin);
    for (int co = 0; co < cout; ++co)
        for (int ci = 0; ci < cin; ++ci)
            for (int t = 0; t < 9; ++t) A[((size_t)t * cout + co) * cin + ci] = h->v[((size_t)co * cin + ci) * 9 + t];
    CHK(e->upload(w, A));
    if (wh && e->un_h16) CHK(upload_h16(e, A, wh));
    if (wx && e->un_x3) CHK(upload_split(e, A, wx));
    h = e->get(name + ".bias", {cout}); if (!h) return DMAD_ERR_STATE;
    CHK(e->upload(b, h->v));
    return 0;
}
int un_dense(dmad_engine* e, const std::string& name, int out, int in, float** w, float** b, h16_t** wh = nullptr, float** wx = nullptr) {
    const HostW* h = e->get(name + ".weight", {out, in}); if (!h) return DMAD_ERR_STATE;
    CHK(e->upload(w, h->v));
    if (wh && e->un_h16) CHK(upload_h16(e, h->v, wh));
    if (wx && e->un_x3) CHK(upload_split(e, h->v, wx));
    h = e->get(name + ".bias", {out}); if (!h) return DMAD_ERR_STATE;
    CHK(e->upload(b, h->v));
    return 0;
}
int un_load_op(dmad_engine* e, const std::string& p, dmad_engine::UnOp& o) {
    if (o.kind == 0) {
        const HostW* h = e->get(p + ".weight", {o.cout, 1, 3, 3}); if (!h) return DMAD_ERR_STATE;
        CHK(e->upload(&o.w1, h->v));
        h = e->get(p + ".bias", {o.cout}); if (!h) return DMAD_ERR_STATE;
        CHK(e->upload(&o.b1, h->v));
    } else if (o.kind == 1) {
        CHK(un_dense(e, p + ".in_layers.0", o.cin, 1, &o.gn1w, &o.gn1b));
        CHK(un_conv3(e, p + ".in_layers.2", o.cout, o.cin, &o.w1, &o.b1, &o.w1h, &o.w1x));
        CHK(un_dense(e, p + ".emb_layers.1", 2 * o.cout, kUnTE, &o.embw, &o.embb));
        CHK(un_dense(e, p + ".out_layers.0", o.cout, 1, &o.gn2w, &o.gn2b));
        CHK(un_conv3(e, p + ".out_layers.3", o.cout, o.cout, &o.w2, &o.b2, &o.w2h, &o.w2x));
        if (o.cin != o.cout) CHK(un_dense(e, p + ".skip_connection", o.cout, o.cin, &o.skw, &o.skb, &o.skwh, &o.skwx));
        o.ss_off = e->un_ss_total;
        e->un_ss_total += (size_t)2 * o.cout;
    } else if (o.kind == 2) {
        CHK(un_dense(e, p + ".norm", o.cin, 1, &o.gn1w, &o.gn1b));
        CHK(un_dense(e, p + ".qkv", 3 * o.cin, o.cin, &o.w1, &o.b1, &o.w1h, &o.w1x));
        CHK(un_dense(e, p + ".proj_out", o.cin, o.cin, &o.w2, &o.b2, &o.w2h, &o.w2x));
    } else if (o.kind == 3) {
        CHK(un_conv3(e, p + ".op", o.cout, o.cin, &o.w1, &o.b1, &o.w1h, &o.w1x));
    } else {
        CHK(un_conv3(e, p + ".conv", o.cout, o.cin, &o.w1, &o.b1, &o.w1h, &o.w1x));
    }
    return 0;
}

int finalize_unet(dmad_engine* e) {
    typedef dmad_engine::UnOp Op;
    auto mk = [](int kind, int cin, int cout) { Op o; o.kind = kind; o.cin = cin; o.cout = cout; return o; };
    // enumerate the modules exactly as UNetModel.__init__ builds them (unet.py:338-421)
    std::vector<int> chans{kUnMC};
    int ch = kUnMC, ds = 1, hw = 1024;
    e->un_in.push_back({mk(0, 1, kUnMC)});
    e->un_hs_ch = {kUnMC}; e->un_hs_hw = {1024};
    for (int level = 0; level < 4; ++level) {
        for (int r = 0; r < kUnRes; ++r) {
            std::vector<Op> blk{mk(1, ch, kUnMult[level] * kUnMC)};
            ch = kUnMult[level] * kUnMC;
            if (un_attn_at(ds)) blk.push_back(mk(2, ch, ch));
            e->un_in.push_back(blk);
            chans.push_back(ch); e->un_hs_ch.push_back(ch); e->un_hs_hw.push_back(hw);
        }
        if (level != 3) {
            e->un_in.push_back({mk(3, ch, ch)});
            ds *= 2; hw /= 4;
            chans.push_back(ch); e->un_hs_ch.push_back(ch); e->un_hs_hw.push_back(hw);
        }
    }
    e->un_mid = {mk(1, ch, ch), mk(2, ch, ch), mk(1, ch, ch)};
    for (int level = 3; level >= 0; --level)
        for (int i = 0; i <= kUnRes; ++i) {
            std::vector<Op> blk{mk(1, ch + chans.back(), kUnMC * kUnMult[level])};
            chans.pop_back();
            ch = kUnMC * kUnMult[level];
            if (un_attn_at(ds)) blk.push_back(mk(2, ch, ch));
            if (level && i == kUnRes) { blk.push_back(mk(4, ch, ch)); ds /= 2; }
            e->un_out.push_back(blk);
        }
    for (size_t i = 0; i < e->un_in.size(); ++i)
        for (size_t j = 0; j < e->un_in[i].size(); ++j)
            CHK(un_load_op(e, "un.input_blocks." + std::to_string(i) + "." + std::to_string(j), e->un_in[i][j]));
    for (size_t j = 0; j < e->un_mid.size(); ++j) CHK(un_load_op(e, "un.middle_block." + std::to_string(j), e->un_mid[j]));
    for (size_t i = 0; i < e->un_out.size(); ++i)
        for (size_t j = 0; j < e->un_out[i].size(); ++j)
            CHK(un_load_op(e, "un.output_blocks." + std::to_string(i) + "." + std::to_string(j), e->un_out[i][j]));
    CHK(un_dense(e, "un.time_embed.0", kUnTE, kUnMC, &e->un_te0w, &e->un_te0b));
    CHK(un_dense(e, "un.time_embed.2", kUnTE, kUnTE, &e->un_te2w, &e->un_te2b));
    CHK(un_dense(e, "un.out.0", kUnMC, 1, &e->un_outgw, &e->un_outgb));
    CHK(un_conv3(e, "un.out.2", 1, kUnMC, &e->un_outw, &e->un_outb));
    const size_t B = (size_t)e->maxB;
    for (size_t i = 0; i < e->un_hs_ch.size(); ++i) {
        float* p = nullptr;
        CHK(e->alloc(&p, B * e->un_hs_hw[i] * e->un_hs_ch[i]));
        e->un_hs.push_back(p);
    }
    for (int i = 0; i < 8; ++i) CHK(e->alloc(&e->un_buf[i], B * 1024 * 384));     // largest map: 32x32 x (256 + 128) concat
    CHK(e->alloc(&e->un_eps, B * 1024));
    if (e->un_h16) {
        for (size_t i = 0; i < e->un_hs_ch.size(); ++i) {
            h16_t* p = nullptr;
            CHK(e->alloc(&p, B * e->un_hs_hw[i] * e->un_hs_ch[i]));
            e->un_hs16.push_back(p);
        }
        for (int i = 0; i < 3; ++i) CHK(e->alloc(&e->un_buf16[i], B * 1024 * 384));
        // statistics slabs: [pixels / 64][channels / 4][2] floats = 1 / 32 float per map value
        for (int i = 0; i < 3; ++i) CHK(e->alloc(&e->un_st_buf[i], B * 1024 * 384 / 32));
        CHK(e->alloc(&e->un_st_t2, B * 1024 * 256 / 32));
        for (size_t i = 0; i < e->un_hs_ch.size(); ++i) {
            float* p = nullptr;
            CHK(e->alloc(&p, B * e->un_hs_hw[i] * e->un_hs_ch[i] / 32 + 64));
            e->un_st_hs.push_back(p);
        }
        CHK(e->alloc(&e->un_t1h, B * 1024 * 384));
        CHK(e->alloc(&e->un_t2h, B * 1024 * 256));
        CHK(e->alloc(&e->un_uph, B * 1024 * 256));
        CHK(e->alloc(&e->un_atth, B * 256 * 256));
        CHK(e->alloc(&e->un_qkvh, B * 256 * 768));
        if (int r = gemm_h16_configure()) return fail(DMAD_ERR_HIP, "hipFuncSetAttribute(max dynamic LDS, f16 conv GEMM) failed: %d", r);
    }
    if (e->un_x3) if (int r = gemm_x3_configure()) return fail(DMAD_ERR_HIP, "hipFuncSetAttribute(max dynamic LDS, split-f16 tier) failed: %d", r);
    CHK(e->alloc(&e->un_ss_table, (size_t)(kUnSsSteps + 1) * e->un_ss_total));
    e->un_ss_have.assign(kUnSsSteps, 0);
    e->un_t = -1;
    CHK(e->alloc(&e->un_temb, kUnMC)); CHK(e->alloc(&e->un_emb1, kUnTE)); CHK(e->alloc(&e->un_emb, kUnTE)); CHK(e->alloc(&e->un_semb, kUnTE));
    return 0;
}

// emb = time_embed(timestep_embedding(t)) (unet.py:466, nn.py:103-121), SiLU(emb), and every ResBlock's (scale, shift) row
int unet_prepare_step(dmad_engine* e, int t, hipStream_t s) {
    if (e->un_t == t) return 0;
    const int slot = t < kUnSsSteps ? t : kUnSsSteps;
    float* row = e->un_ss_table + (size_t)slot * e->un_ss_total;
    e->un_ss_cur = row;
    e->un_t = t;
    if (slot < kUnSsSteps && e->un_ss_have[slot]) return 0;
    float te[kUnMC];
    const float a = (float)(-log(10000.0));
    for (int i = 0; i < kUnMC / 2; ++i) {
        const float f = expf(a * (float)i / (float)(kUnMC / 2));
        const float arg = (float)t * f;
        te[i] = cosf(arg);
        te[kUnMC / 2 + i] = sinf(arg);
    }
    static_assert(kUnMC == 128, "launch_store_vec128 carries 128 floats");
    launch_store_vec128(te, e->un_temb, s); // as a kernel argument: no host buffer to keep alive, no synchronisation
    launch_gemm_f32(plain_gemm(e->un_te0w, e->un_temb, e->un_emb1, nullptr, e->un_te0b, kUnTE, kUnMC, 1, kUnTE, kUnMC, 0), s);
    launch_silu(e->un_emb1, e->un_emb1, kUnTE, s);
    launch_gemm_f32(plain_gemm(e->un_te2w, e->un_emb1, e->un_emb, nullptr, e->un_te2b, kUnTE, kUnTE, 1, kUnTE, kUnTE, 0), s);
    launch_silu(e->un_emb, e->un_semb, kUnTE, s);
    auto each = [&](dmad_engine::UnOp& o) {
        if (o.kind == 1)
            launch_gemm_f32(plain_gemm(o.embw, e->un_semb, row + o.ss_off, nullptr, o.embb, 2 * o.cout, kUnTE, 1, 2 * o.cout, kUnTE, 0), s);
    };
    for (auto& b : e->un_in) for (auto& o : b) each(o);
    for (auto& o : e->un_mid) each(o);
    for (auto& b : e->un_out) for (auto& o : b) each(o);
    if (slot < kUnSsSteps) e->un_ss_have[slot] = 1;
    return 0;
}

GemmF32Args un_conv_args(const float* A, const float* bias, const float* X, float* C, int cout, int cin, int taps, int B, int H, int stride,
                         const float* res) {
    GemmF32Args g{};
    const int Ho = (H - 1) / (stride > 1 ? stride : 1) + 1;
    g.A = A; g.X = X; g.C = C; g.scale = nullptr; g.shift = bias; g.M = cout; g.K = cin; g.taps = taps; g.ldc = cout; g.relu = 0;
    g.N = (long)B * Ho * Ho; g.mode = 2; g.H = H; g.W = H; g.Cin = cin; g.ldx = cin; g.stride = stride; g.res = res;
    return g;
}

// a dense layer over NHWC rows as a 1x1 conv (mode 2: the form both the fp32 and the split-f16 launch paths serve for any M % 128 == 0)
GemmF32Args plain_conv1x1(const float* A, const float* bias, const float* X, float* C, int M, int K, long N) {
    GemmF32Args g{};
    g.A = A; g.X = X; g.C = C; g.shift = bias; g.M = M; g.K = K; g.taps = 1; g.ldc = M; g.N = N; g.mode = 2; g.H = 1; g.W = 1; g.Cin = K; g.ldx = K; g.stride = 1;
    return g;
}

const float* gn_fail(int HW, int C) { fail(DMAD_ERR_STATE, "GroupNorm: no kernel for a %d-pixel x %d-channel map", HW, C); return nullptr; }

// applies one module; `in` [B][H*H][cin] -> returns the buffer holding [B][Ho*Ho][cout].  `dst`: where the result must
// land (a saved-skip buffer) or nullptr (take a rotating work buffer).
// `in2` != nullptr (ResBlocks of the output path only): the module's input is th.cat([in, in2], dim=1) (unet.py:473), `in` holding
// c1 channels and `in2` the rest — GroupNorm and the 1x1 skip conv read the two parts in place, nothing is concatenated.
// x3: the MIDDLE tier — the same fp32 pipeline (fp32 maps, GroupNorm, softmax, residual sums) with every conv / 1x1 on split-f16 operands
// (three f16 MFMAs per product, ~22 significant bits, gemm_x3_kernel): GroupNorm writes its output in the split format, the maps a GEMM
// reads without a GroupNorm in between (the block input of a 1x1 skip conv, of a Downsample / Upsample conv, the attention output) are
// converted by one elementwise pass.
const float* unet_apply(dmad_engine* e, const dmad_engine::UnOp& o, const float* in, int B, int& H, float* dst, int& rot, hipStream_t s,
                        const float* in2 = nullptr, int c1 = 0, bool x3 = false) {
    float *T1 = e->un_buf[3], *T2 = e->un_buf[4], *SK = e->un_buf[5], *QKV = e->un_buf[6], *ATT = e->un_buf[7];
    auto next = [&]() { float* p = e->un_buf[rot]; rot = (rot + 1) % 3; if (p == in) { p = e->un_buf[rot]; rot = (rot + 1) % 3; } return p; };
    float* out = dst ? dst : next();
    const long nref = (long)e->maxB * H * H;
    auto gemm = [&](GemmF32Args g, const float* wx, long nr) {             // one conv / 1x1 on this pass's tier
        if (x3) { g.A = wx; g.x3 = 1; launch_gemm_f32(g, s); }
        else launch_gemm_f32(g, s, e->slab, e->slab_floats, nr);
    };
    if (o.kind == 1) {                      // ResBlock._forward, unet.py:186-199
        if (in2 && o.cin == o.cout) { fail(DMAD_ERR_STATE, "a concatenated input needs the ResBlock's skip conv"); return nullptr; }
        if (launch_groupnorm_nhwc(in, o.gn1w, o.gn1b, nullptr, 1, T1, B, H * H, o.cin, s, in2, c1, nullptr, nullptr, nullptr, x3)) return gn_fail(H * H, o.cin);
        gemm(un_conv_args(o.w1, o.b1, T1, T2, o.cout, o.cin, 9, B, H, 1, nullptr), o.w1x, nref);
        if (launch_groupnorm_nhwc(T2, o.gn2w, o.gn2b, e->un_ss_cur + o.ss_off, 1, T1, B, H * H, o.cout, s, nullptr, 0, nullptr, nullptr, nullptr, x3)) return gn_fail(H * H, o.cout);
        const float* skip = in;
        if (o.cin != o.cout) {
            const float *sin = in, *sin2 = in2;
            if (x3) {                       // the block input(s) as split-format operands (QKV / ATT are free inside a ResBlock)
                const int ca = in2 ? c1 : o.cin;
                launch_scale(in, 1.f, QKV, (long)B * H * H * ca, s, true);
                sin = QKV;
                if (in2) { launch_scale(in2, 1.f, ATT, (long)B * H * H * (o.cin - c1), s, true); sin2 = ATT; }
            }
            GemmF32Args g = un_conv_args(o.skw, o.skb, sin, SK, o.cout, o.cin, 1, B, H, 1, nullptr);
            if (in2) { g.ldx = c1; g.X2 = sin2; g.ksplit = c1; g.ldx2 = o.cin - c1; }
            gemm(g, o.skwx, nref);
            skip = SK;
        }
        gemm(un_conv_args(o.w2, o.b2, T1, out, o.cout, o.cout, 9, B, H, 1, skip), o.w2x, nref);
    } else if (o.kind == 2) {               // AttentionBlock._forward + QKVAttention, unet.py:225-258
        const int C = o.cin, T = H * H;
        if (launch_groupnorm_nhwc(in, o.gn1w, o.gn1b, nullptr, 0, T1, B, T, C, s, nullptr, 0, nullptr, nullptr, nullptr, x3)) return gn_fail(T, C);
        gemm(x3 ? plain_conv1x1(o.w1, o.b1, T1, QKV, 3 * C, C, (long)B * T) : plain_gemm(o.w1, T1, QKV, nullptr, o.b1, 3 * C, C, (long)B * T, 3 * C, C, 0), o.w1x, nref);
        if (int rc = launch_qkv_attention(QKV, ATT, B, T, kUnHeads, s, nullptr, x3 ? 1 : 0)) { fail(rc > 0 ? DMAD_ERR_HIP : DMAD_ERR_STATE, "UNet attention (T = %d): %s", T, rc > 0 ? hipGetErrorString((hipError_t)rc) : "unsupported map size"); return nullptr; }      // (x3: the output straight in the split format)
        GemmF32Args g = x3 ? plain_conv1x1(o.w2, o.b2, ATT, out, C, C, (long)B * T) : plain_gemm(o.w2, ATT, out, nullptr, o.b2, C, C, (long)B * T, C, C, 0);
        g.res = in;
        gemm(g, o.w2x, nref);
    } else if (o.kind == 3) {               // Downsample: conv 3x3 stride 2, unet.py:82-111
        const float* xin = in;
        if (x3) { launch_scale(in, 1.f, T1, (long)B * H * H * o.cin, s, true); xin = T1; }
        gemm(un_conv_args(o.w1, o.b1, xin, out, o.cout, o.cin, 9, B, H, 2, nullptr), o.w1x, nref / 4);
        H /= 2;
    } else if (o.kind == 4) {               // Upsample: nearest x2 + conv 3x3, unet.py:49-79
        launch_upsample2x_nhwc(in, T1, B, H, H, o.cin, s);
        H *= 2;
        if (x3) launch_scale(T1, 1.f, T1, (long)B * H * H * o.cin, s, true);
        gemm(un_conv_args(o.w1, o.b1, T1, out, o.cout, o.cin, 9, B, H, 1, nullptr), o.w1x, nref * 4);
    } else {
        if (launch_conv1ch_3x3(in, o.w1, o.b1, out, B, o.cout, s)) { fail(DMAD_ERR_STATE, "input conv: %d output channels > 128", o.cout); return nullptr; }
    }
    return out;
}

// ---- the same network on the 16-bit tier: every conv / 1x1 through gemm_h16 (f16 operands, fp32 accumulate); GroupNorm statistics,
// softmax and the bias / residual sums in fp32; the hidden state itself exists as f16 maps ONLY (a block output costs 2 bytes per
// value to write and 2 to read back as the next residual, against 4 + 2 and 4 with an fp32 copy beside it).  UMap::f of a block
// output is only the identity of its buffer slot on this tier (never written or read); the network input has f alone.
struct UMap { const float* f; const h16_t* h; const float* st; };      // st: the map's GroupNorm statistics slab (nullptr: none, e.g. maps of fewer than 64 pixels)

GemmH16Args un_h16_args(const h16_t* A, const float* bias, const h16_t* X, float* C, h16_t* C16, int cout, int cin, int taps, int B, int H,
                        int stride, const h16_t* res16, float* stats = nullptr) {
    GemmH16Args g{};
    const int Ho = (H - 1) / (stride > 1 ? stride : 1) + 1;
    g.A = A; g.X = X; g.C = C; g.C16 = C16; g.shift = bias; g.res16 = res16; g.M = cout; g.K = cin; g.taps = taps; g.ldc = cout;
    g.N = (long)B * Ho * Ho; g.H = H; g.W = H; g.ldx = cin; g.stride = stride;
    g.stats = Ho * Ho >= 16 ? stats : nullptr;          // a statistics block must lie inside one sample: 64 pixels, or 16 on the 4 x 4 maps
    g.stats_px = Ho * Ho >= 64 ? 64 : 16;
    return g;
}

// GroupNorm of the 16-bit tier: the one-pass kernel when every part of the input carries its statistics slab, the two-pass ones otherwise
int un_groupnorm16(UMap in, UMap in2, int c1, const float* gw, const float* gb, const float* ss, int silu, h16_t* y16, float* y32, int B, int HW,
                   int C, hipStream_t s) {
    static const bool fused = []() { const char* v = getenv("DMAD_GN_FUSED"); return !(v && v[0] == '0'); }();     // A/B switch
    if (fused && in.h && in.st && (!in2.h || in2.st) && HW >= 16 &&
        launch_groupnorm16_apply(in.h, in.st, in2.h, in2.st, c1, gw, gb, ss, silu, y16, y32, B, HW, C, s) == 0)
        return 0;
    return launch_groupnorm_nhwc(in.f, gw, gb, ss, silu, y32, B, HW, C, s, in2.f, c1, y16, in.h, in2.h);
}

bool unet_apply_h16(dmad_engine* e, const dmad_engine::UnOp& o, UMap in, int B, int& H, float* dstf, h16_t* dsth, float* dstst, int& rot, hipStream_t s,
                    UMap* result, UMap in2 = UMap{nullptr, nullptr, nullptr}, int c1 = 0) {
    h16_t* SK16 = (h16_t*)e->un_buf[5];        // the skip conv's output, f16 (the fp32 tier's buffer, reused)
    h16_t *T1h = e->un_t1h, *T2h = e->un_t2h, *ATTh = e->un_atth;
    float* outf = dstf;
    h16_t* outh = dsth;
    float* outst = dstst;
    if (!outf) {
        int r = rot; rot = (rot + 1) % 3;
        if (e->un_buf[r] == in.f) { r = rot; rot = (rot + 1) % 3; }
        outf = e->un_buf[r]; outh = e->un_buf16[r]; outst = e->un_st_buf[r];
    }
    int Hout = H;
    const UMap none{nullptr, nullptr, nullptr};
    if (o.kind == 1) {                      // ResBlock._forward, unet.py:186-199
        if (in2.f && o.cin == o.cout) { fail(DMAD_ERR_STATE, "a concatenated input needs the ResBlock's skip conv"); return false; }
        // GroupNorm reads the f16 maps (statistics from the producing GEMM's epilogue where there is one); the in_layers conv writes f16 only
        if (un_groupnorm16(in, in2, c1, o.gn1w, o.gn1b, nullptr, 1, T1h, nullptr, B, H * H, o.cin, s)) { gn_fail(H * H, o.cin); return false; }
        launch_gemm_h16(un_h16_args(o.w1h, o.b1, T1h, nullptr, T2h, o.cout, o.cin, 9, B, H, 1, nullptr, e->un_st_t2), s);
        const UMap t2{nullptr, T2h, H * H >= 16 ? e->un_st_t2 : nullptr};
        if (un_groupnorm16(t2, none, 0, o.gn2w, o.gn2b, e->un_ss_cur + o.ss_off, 1, T1h, nullptr, B, H * H, o.cout, s)) { gn_fail(H * H, o.cout); return false; }
        const h16_t* skip = in.h;
        if (o.cin != o.cout) {
            GemmH16Args g = un_h16_args(o.skwh, o.skb, in.h, nullptr, SK16, o.cout, o.cin, 1, B, H, 1, nullptr);
            if (in2.f) { g.ldx = c1; g.X2 = in2.h; g.ksplit = c1; g.ldx2 = o.cin - c1; }
            launch_gemm_h16(g, s);
            skip = SK16;
        }
        launch_gemm_h16(un_h16_args(o.w2h, o.b2, T1h, nullptr, outh, o.cout, o.cout, 9, B, H, 1, skip, outst), s);
    } else if (o.kind == 2) {               // AttentionBlock._forward + QKVAttention, unet.py:225-258
        const int C = o.cin, T = H * H;
        if (un_groupnorm16(in, none, 0, o.gn1w, o.gn1b, nullptr, 0, T1h, nullptr, B, T, C, s)) { gn_fail(T, C); return false; }
        if ((long)T * C > 256l * 256) { fail(DMAD_ERR_STATE, "UNet attention: %d tokens x %d channels exceed the f16 qkv buffer", T, C); return false; }
        launch_gemm_h16(un_h16_args(o.w1h, o.b1, T1h, nullptr, e->un_qkvh, 3 * C, C, 1, B, H, 1, nullptr), s);      // qkv straight to f16
        if (int rc = launch_qkv_attention_h16(e->un_qkvh, ATTh, B, T, kUnHeads, s)) { fail(rc > 0 ? DMAD_ERR_HIP : DMAD_ERR_STATE, "UNet attention (T = %d): %s", T, rc > 0 ? hipGetErrorString((hipError_t)rc) : "unsupported map size"); return false; }
        launch_gemm_h16(un_h16_args(o.w2h, o.b2, ATTh, nullptr, outh, C, C, 1, B, H, 1, in.h, outst), s);
    } else if (o.kind == 3) {               // Downsample: conv 3x3 stride 2, unet.py:82-111
        launch_gemm_h16(un_h16_args(o.w1h, o.b1, in.h, nullptr, outh, o.cout, o.cin, 9, B, H, 2, nullptr, outst), s);
        H /= 2;
        Hout = H;
    } else if (o.kind == 4) {               // Upsample: nearest x2 + conv 3x3, unet.py:49-79
        H *= 2;
        Hout = H;
        GemmH16Args g = un_h16_args(o.w1h, o.b1, in.h, nullptr, outh, o.cout, o.cin, 9, B, H, 1, nullptr, outst);
        g.up2 = 1;                              // the conv reads the half-resolution map through the upsampling where the kernel can
        if (!gemm_h16_fuses_up2(g)) {
            launch_upsample2x_nhwc_h16(in.h, e->un_uph, B, H / 2, H / 2, o.cin, s);
            g.up2 = 0;
            g.X = e->un_uph;
        }
        launch_gemm_h16(g, s);
    } else {                                // input conv 1 -> 128 (direct kernel, fp32 arithmetic): the f16 map and its statistics only
        if (launch_conv1ch_3x3(in.f, o.w1, o.b1, nullptr, B, o.cout, s, outh, (o.cout & 3) ? nullptr : outst)) { fail(DMAD_ERR_STATE, "input conv: %d output channels > 128", o.cout); return false; }
        if (o.cout & 3) outst = nullptr;
    }
    *result = UMap{outf, outh, Hout * Hout >= 16 ? outst : nullptr};
    return true;
}

// eps = UNetModel.forward(x, t * ones)  (unet.py:453-477): x, eps [B][32][32]
// h16 = 2: the split-f16 middle tier (fp32 pipeline, every conv on split-f16 operands: fp32-grade at several times the fp32 matrix rate).
// h16 < 0: the tier of the map-returning entry points (dmad_unet_eps / dmad_unet_p_sample / dmad_spec_query_logits): the 16-bit tier in
// DMAD_MODE_FAST (and in DMAD_MODE_EXACT_VOTES when dmad_set_waveform_tier chose the 16-bit tier), in DMAD_MODE_EXACT_VOTES otherwise the
// tier dmad_set_waveform_tier selects — the split-f16 tier by default (fp32-grade, 2.2 x the fp32 rate), the exact-fp32 UNet on request
// and in DMAD_MODE_FP32 — only the spec-domain vote loop has a recheck, so it alone runs the 16-bit tier by default (it passes h16 = 1);
// 0 / 1 / 2: explicit
int unet_eps(dmad_engine* e, const float* x, int t, int B, float* eps, hipStream_t s, int h16 = -1) {
    if (!e->un_final) return fail(DMAD_ERR_STATE, "UNet weights are not finalised (dmad_load_weight + dmad_finalize_weights)");
    if (B < 1 || B > e->maxB) return fail(DMAD_ERR_STATE, "batch %d outside [1, max_batch=%d]", B, e->maxB);
    if (t < 0) return fail(DMAD_ERR_INVALID, "diffusion step %d < 0", t);
    CHK(unet_prepare_step(e, t, s));
    if (h16 < 0) {      // the map-returning surfaces follow dmad_set_waveform_tier like the waveform-returning ones: split-f16 by default on exact-vote engines
        if (e->un_h16 && (e->mode == DMAD_MODE_FAST || (e->mode == DMAD_MODE_EXACT_VOTES && e->wave_tier == PATH_DEFAULT))) h16 = 1;
        else if (e->un_x3 && e->mode == DMAD_MODE_EXACT_VOTES && e->wave_tier == PATH_X3) h16 = 2;
        else h16 = 0;
    }
    if (h16 == 1 && !e->un_h16) return fail(DMAD_ERR_STATE, "this engine has no 16-bit UNet tier (DMAD_FP32 precision)");
    if (h16 == 1) {
        int H = 32, rot = 0;
        UMap h{x, nullptr, nullptr};
        std::vector<const float*> hs_st(e->un_hs.size(), nullptr);      // the statistics slab each saved map ended up with (16-pixel blocks on the 4x4 maps)
        for (size_t i = 0; i < e->un_in.size(); ++i)
            for (size_t j = 0; j < e->un_in[i].size(); ++j) {
                const bool save = j + 1 == e->un_in[i].size();
                if (!unet_apply_h16(e, e->un_in[i][j], h, B, H, save ? e->un_hs[i] : nullptr, save ? e->un_hs16[i] : nullptr, save ? e->un_st_hs[i] : nullptr, rot, s, &h)) return DMAD_ERR_STATE;
                if (save) hs_st[i] = h.st;
            }
        for (auto& o : e->un_mid) if (!unet_apply_h16(e, o, h, B, H, nullptr, nullptr, nullptr, rot, s, &h)) return DMAD_ERR_STATE;
        size_t top = e->un_hs.size();
        for (auto& blk : e->un_out) {
            --top;
            const int c1 = blk[0].cin - e->un_hs_ch[top];
            const UMap hs{e->un_hs[top], e->un_hs16[top], hs_st[top]};
            for (size_t j = 0; j < blk.size(); ++j) {
                const bool ok = j == 0 ? unet_apply_h16(e, blk[0], h, B, H, nullptr, nullptr, nullptr, rot, s, &h, hs, c1) : unet_apply_h16(e, blk[j], h, B, H, nullptr, nullptr, nullptr, rot, s, &h);
                if (!ok) return DMAD_ERR_STATE;
            }
        }
        if (un_groupnorm16(h, UMap{nullptr, nullptr, nullptr}, 0, e->un_outgw, e->un_outgb, nullptr, 1, e->un_t1h, nullptr, B, 1024, kUnMC, s)) { gn_fail(1024, kUnMC); return DMAD_ERR_STATE; }
        launch_conv3x3_c128_to1_h16(e->un_t1h, e->un_outw, e->un_outb, eps, B, s);       // (operands f16, fp32 accumulate, like the tier's GEMMs)
        LASTCHK();
        return 0;
    }
    const bool x3 = h16 == 2;
    if (x3 && !e->un_x3) return fail(DMAD_ERR_STATE, "this engine has no split-f16 UNet tier (it needs DMAD_EXACT precision)");
    int H = 32, rot = 0;
    const float* h = x;
    for (size_t i = 0; i < e->un_in.size(); ++i)
        for (size_t j = 0; j < e->un_in[i].size(); ++j)
            if (!(h = unet_apply(e, e->un_in[i][j], h, B, H, j + 1 == e->un_in[i].size() ? e->un_hs[i] : nullptr, rot, s, nullptr, 0, x3))) return DMAD_ERR_STATE;
    for (auto& o : e->un_mid) if (!(h = unet_apply(e, o, h, B, H, nullptr, rot, s, nullptr, 0, x3))) return DMAD_ERR_STATE;
    size_t top = e->un_hs.size();
    for (auto& blk : e->un_out) {
        --top;
        const int c1 = blk[0].cin - e->un_hs_ch[top];          // th.cat([h, hs.pop()], dim=1): h carries c1 channels, the saved map the rest
        const float* hs = e->un_hs[top];
        for (size_t j = 0; j < blk.size(); ++j) {
            h = j == 0 ? unet_apply(e, blk[0], h, B, H, nullptr, rot, s, hs, c1, x3) : unet_apply(e, blk[j], h, B, H, nullptr, rot, s, nullptr, 0, x3);
            if (!h) return DMAD_ERR_STATE;
        }
    }
    if (launch_groupnorm_nhwc(h, e->un_outgw, e->un_outgb, nullptr, 1, e->un_buf[3], B, 1024, kUnMC, s)) { gn_fail(1024, kUnMC); return DMAD_ERR_STATE; }
    static_assert(kUnMC == 128, "launch_conv3x3_c128_to1 is the 128-channel output layer");
    launch_conv3x3_c128_to1(e->un_buf[3], e->un_outw, e->un_outb, eps, B, s);       // (the 128 -> 1 output conv: exact fp32 on every tier but the 16-bit one)
    LASTCHK();
    return 0;
}

int ensure_embed(dmad_engine* e, int t, hipStream_t s) {
    if (e->emb_t == t) return 0;
    launch_embed_table((float)t, e->fc1w, e->fc1b, e->fc2w, e->fc2b, e->fctw, e->fctb, e->emb_table, e->emb2, e->bf16 ? e->b2 : nullptr,
                       e->bf16 ? e->epi_c : nullptr, e->NL, s);
    e->emb_t = t;
    return 0;
}

GemmF32Args plain_gemm(const float* A, const float* X, float* C, const float* scale, const float* shift, int M, int K, long N,
                       int ldc, long ldx, int relu) {
    GemmF32Args g{};
    g.A = A; g.X = X; g.C = C; g.scale = scale; g.shift = shift;
    g.M = M; g.K = K; g.taps = 1; g.ldc = ldc; g.relu = relu; g.N = N; g.mode = 0;
    g.rows_per_batch = N > 0 ? N : 1; g.batch_stride = 0; g.row_stride = ldx; g.tap_stride = 0;
    return g;
}

// exact32: evaluate on the exact-fp32 path (the only one of a DMAD_FP32 engine; DMAD_MODE_FP32 and the recheck pass of a
// DMAD_EXACT engine); batches larger than the fp32 workspace are walked in chunks of maxB32 clips
int wavenet_eps(dmad_engine* e, const float* x_t, int t, int B, float* eps, hipStream_t s, int path = PATH_DEFAULT) {
    if (!e->wn_final) return fail(DMAD_ERR_STATE, "WaveNet weights are not finalised (dmad_load_weight + dmad_finalize_weights)");
    if (B < 1 || B > e->maxB) return fail(DMAD_ERR_STATE, "batch %d outside [1, max_batch=%d]", B, e->maxB);
    if (t < 0) return fail(DMAD_ERR_INVALID, "diffusion step %d < 0", t);
    CHK(ensure_embed(e, t, s));
    const int L = e->L, LP = e->LP, NL = e->NL;
    const bool use32 = !e->bf16 || (e->f32 && (path != PATH_DEFAULT || e->mode == DMAD_MODE_FP32));
    const bool x3 = use32 && path == PATH_X3 && e->wdil_x3;     // fp32 pipeline on split-f16 operands (three MFMAs per product)
    if (use32 && B > e->maxB32) {
        for (int b0 = 0; b0 < B; b0 += e->maxB32) {
            const int bb = B - b0 < e->maxB32 ? B - b0 : e->maxB32;
            CHK(wavenet_eps(e, x_t + (size_t)b0 * L, t, bb, eps + (size_t)b0 * L, s, x3 ? PATH_X3 : PATH_FP32));
        }
        return 0;
    }
    if (!use32) {
        launch_wn_init_bf16(x_t, e->init_w, e->init_b, e->emb_table, e->hA, B, L, LP, e->f16, s);
        for (int n = 0; n < NL; ++n) {
            WnLayerArgs a{};
            a.hin = (n & 1) ? e->hB : e->hA;
            a.hout = (n & 1) ? e->hA : e->hB;
            a.gout = e->gstore + (size_t)n * B * L * kC;
            a.w1p = e->w1p + (size_t)n * 24 * 512 * 32;
            a.w2p = e->w2p + (size_t)n * 8 * 256 * 32;
            a.b1 = e->b1p + (size_t)n * 512;
            a.epi_c = e->epi_c + (size_t)n * 256;
            a.dilation = 1 << (n % e->cfg.dilation_cycle);
            a.L = L; a.LP = LP; a.last = (n == NL - 1); a.npos = (long)B * L;
            const bool timed = e->prof_on && !a.last && e->prof_used + 2 <= e->prof_ev.size();
            if (timed) (void)hipEventRecord(e->prof_ev[e->prof_used++], s);
            launch_wn_layer_bf16_p(a, B, e->f16, s);
            if (timed) (void)hipEventRecord(e->prof_ev[e->prof_used++], s);
        }
        WnFinalArgs f{};
        f.g = e->gstore; f.wsp = e->wsp; f.wf0p = e->wf0p; f.bskip_sum = e->bskip_sum; f.bf0 = e->bf0; f.wz = e->wz;
        f.eps = eps; f.bz = e->bz; f.skip_scale = (float)sqrt(1.0 / NL); f.NL = NL; f.B = B; f.L = L;
        const bool timed_f = e->prof_on && e->prof_used_f + 2 <= e->prof_ev_f.size();
        if (timed_f) (void)hipEventRecord(e->prof_ev_f[e->prof_used_f++], s);
        launch_wn_final_bf16_p(f, e->f16, s);
        if (timed_f) (void)hipEventRecord(e->prof_ev_f[e->prof_used_f++], s);
    } else {
        const long N = (long)B * L;
        launch_wn_init_f32(x_t, e->init_w, e->init_b, e->emb_table, e->hA32, B, L, LP, s, x3, x3 && e->diag[4]);
        const size_t slab = (size_t)e->maxB32 * L * 256;          // one gate-output slab per layer
        for (int n = 0; n < NL; ++n) {
            float* hin = (n & 1) ? e->hB32 : e->hA32;
            float* hout = (n & 1) ? e->hA32 : e->hB32;
            const int d = 1 << (n % e->cfg.dilation_cycle);
            const bool last = n == NL - 1;
            float* gout = e->gstore32 + (size_t)n * slab;
            GemmF32Args g{};
            g.A = (x3 ? e->wdil_x3 : e->wdil) + (size_t)n * 3 * 512 * 256; g.X = hin + (size_t)kPad * kC; g.scale = nullptr;
            g.x3 = x3; g.diag = x3 ? e->diag[0] : 0;
            g.shift = e->bdil + (size_t)n * 512; g.M = 512; g.K = 256; g.taps = 3; g.ldc = 512; g.relu = 0; g.N = N; g.mode = 0;
            g.rows_per_batch = L; g.batch_stride = (long)LP * kC; g.row_stride = kC; g.tap_stride = (long)d * kC;
            g.epi = 1; g.C = gout;               // tanh * sigmoid in the epilogue: H never goes to HBM
            launch_gemm_f32(g, s);
            if (last) continue;                  // the last layer's residual output is never consumed (WaveNet.py:131-135)
            // res conv with the residual update in its epilogue; the skip convs run as one GEMM after the loop
            GemmF32Args u = plain_gemm((x3 ? e->wrs_x3 : e->wrs) + (size_t)n * 512 * 256, gout, nullptr, nullptr, e->brs + (size_t)n * 512, 256, 256,
                                       N, 256, 256, 0);
            u.epi = 2; u.res_rows = 256; u.first = 0; u.L = L; u.LP = LP;
            u.hin = hin; u.hout = hout; u.skip = nullptr; u.emb_next = e->emb_table + (size_t)(n + 1) * 256;
            u.x3 = x3; u.diag = x3 ? e->diag[1] : 0;
            launch_gemm_f32(u, s);
        }
        {   // skip = sum_n W_skip_n g_n + sum_n b_skip_n: taps = layers, tap stride = one slab (taps are centred on NL / 2)
            GemmF32Args k{};
            k.A = x3 ? e->wskip_x3 : e->wskip32; k.X = e->gstore32 + (size_t)(NL >> 1) * slab; k.C = e->skip32; k.scale = nullptr;
            k.shift = e->bskip32; k.M = 256; k.K = 256; k.taps = NL; k.ldc = 256; k.relu = 0; k.N = N; k.mode = 0; k.x3 = x3; k.diag = x3 ? e->diag[2] : 0;
            k.rows_per_batch = N; k.batch_stride = 0; k.row_stride = 256; k.tap_stride = (long)slab;
            launch_gemm_f32(k, s);
        }
        launch_scale(e->skip32, (float)sqrt(1.0 / NL), e->g32, N * 256, s, x3);
        GemmF32Args f = plain_gemm(x3 ? e->wf0_x3 : e->wf0, e->g32, e->H32, nullptr, e->bf0, 256, 256, N, 256, 256, 1);
        f.x3 = x3; f.diag = x3 ? e->diag[3] : 0;
        launch_gemm_f32(f, s);
        launch_dot256(e->H32, e->wz, e->bz, eps, N, s);
    }
    LASTCHK();
    return 0;
}

int mel_db(dmad_engine* e, const float* x, int B, float* spec, hipStream_t s, int to_db = 1) {
    if (!e->cfg.with_classifier) return fail(DMAD_ERR_STATE, "engine was created with with_classifier = 0");
    if (B < 1 || B > e->maxB) return fail(DMAD_ERR_STATE, "batch %d outside [1, max_batch=%d]", B, e->maxB);
    const long rows = (long)B * 32;
    launch_mel_pad(x, e->mel_xp, B, e->L, e->LPm, s);
    GemmF32Args g{};
    g.A = e->dftA; g.X = e->mel_xp; g.C = e->dftD; g.scale = nullptr; g.shift = nullptr;
    g.M = kDftM; g.K = 2048; g.taps = 1; g.ldc = kDftLd; g.relu = 0; g.N = rows; g.mode = 0;
    g.rows_per_batch = 32; g.batch_stride = e->LPm; g.row_stride = 512; g.tap_stride = 0;
    launch_gemm_f32(g, s);
    launch_mel_power(e->dftD, e->melP, kDftLd, kMelLd, rows, s);
    launch_gemm_f32(plain_gemm(e->fbA, e->melP, e->melM, nullptr, nullptr, 32, kMelLd, rows, 32, kMelLd, 0), s);
    launch_mel_db(e->melM, spec, B, to_db, s);
    LASTCHK();
    return 0;
}

// h16 = 1: the classifier's 16-bit tier where one is resident (ResNeXt29 on engines with a 16-bit side) — the fast mode's; 2: its
// split-f16 tier (exact-vote engines) — tier 1 of the exact-vote loops; every other caller (dmad_classify, the recheck tiers) gets the
// fp32 matrix cores
int classify(dmad_engine* e, const float* spec, int B, float* logits, hipStream_t s, int h16 = 0) {
    if (!e->cfg.with_classifier) return fail(DMAD_ERR_STATE, "engine was created with with_classifier = 0");
    if (!e->cls_final) return fail(DMAD_ERR_STATE, "classifier weights are not finalised (dmad_load_weight + dmad_finalize_weights)");
    if (B < 1 || B > e->maxB) return fail(DMAD_ERR_STATE, "batch %d outside [1, max_batch=%d]", B, e->maxB);
    if (e->cls_kind == 1) return (h16 == 1 && e->rx_h16) ? classify_resnext_h16(e, spec, B, logits, s)
                                 : (h16 == 2 && e->rx_x3) ? classify_resnext_x3(e, spec, B, logits, s) : classify_resnext(e, spec, B, logits, s);
    float *cur = e->act0, *nxt = e->act1;
    launch_vgg_conv1(spec, e->vconv1w, e->vscale[0], e->vshift[0], cur, B, s);
    int H = 32, cin = 64, li = 1;
    for (int i = 1; i < kVggCfgLen; ++i) {
        const int v = kVggCfg[i];
        if (v < 0) {
            launch_maxpool2_nhwc(cur, nxt, B, H, H, cin, s);
            H >>= 1;
        } else {
            GemmF32Args g{};
            g.A = e->vconvw[li]; g.X = cur; g.C = nxt; g.scale = e->vscale[li]; g.shift = e->vshift[li];
            g.M = v; g.K = cin; g.taps = 9; g.ldc = v; g.relu = 1; g.N = (long)B * H * H; g.mode = 2;
            g.H = H; g.W = H; g.Cin = cin;
            launch_gemm_f32(g, s, e->slab, e->slab_floats, (long)e->maxB * H * H);
            cin = v;
            ++li;
        }
        float* t = cur; cur = nxt; nxt = t;
    }
    const int fin[3] = {512, 4096, 4096}, fout[3] = {4096, 4096, e->cfg.num_classes};
    for (int j = 0; j < 3; ++j) {
        float* dst = (j == 2) ? logits : nxt;
        launch_gemm_f32(plain_gemm(e->vfcw[j], cur, dst, nullptr, e->vfcb[j], fout[j], fin[j], B, fout[j], fin[j], j < 2), s, e->slab,
                        e->slab_floats, (long)e->maxB);
        float* t = cur; cur = nxt; nxt = t;
    }
    LASTCHK();
    return 0;
}

// WaveNet path of the entry points that hand waveforms (or logits of purified waveforms) back — dmad_wavenet_eps, dmad_one_shot,
// dmad_ddpm_step / _purify, dmad_query_logits: the mode's own path, except that an exact-vote engine in DMAD_MODE_EXACT_VOTES serves
// them on the tier dmad_set_waveform_tier selected (default: the split-f16 tier, fp32-grade) — only the vote loop has a recheck
inline int wave_path(const dmad_engine* e) {
    if (!(e->bf16 && e->f32) || e->mode != DMAD_MODE_EXACT_VOTES) return PATH_DEFAULT;
    return e->wave_tier;
}

// the classifier tier of a vote loop's FIRST pass (and of the mode-default paths): the 16-bit tier (ResNeXt29) in DMAD_MODE_FAST only.
// Round 5, measured on the calibrated stand-in (profiles/r05b_resnext29_error_attribution.json): the f16 classifier's leader-difference
// error is 0.08-0.16 against 0.016-0.030 for the f16 WaveNet in front of the fp32 classifier — a bound that covered it would send a
// quarter of the samples to the recheck tiers, so the exact-vote mode keeps the classifier on the fp32 matrix cores in every tier.
// It runs the classifier's SPLIT-F16 tier there (fp32-grade: its error, ~1e-4, disappears under the WaveNet's): a third of the fp32 tier's time.
inline int cls_tier(const dmad_engine* e) {
    if (e->mode == DMAD_MODE_FAST) return e->rx_h16 ? 1 : 0;
    return (e->mode == DMAD_MODE_EXACT_VOTES && e->rx_x3 && e->cls_kind == 1) ? 2 : 0;
}

// ... and of a recheck tier: the split-f16 WaveNet tier is paired with the classifier's split-f16 tier (ResNeXt29; both fp32-grade, their
// errors add up to ~3e-4 under tau2 = 1e-3 — the fp32 ResNeXt29 at recheck batch sizes cost 0.8 ms per sample, 45 % on top of the
// WaveNet's), the exact-fp32 WaveNet tier with the fp32 classifier: what reaches tier 3 is the fp32 path bit for bit
inline int cls_tier_of_path(const dmad_engine* e, int path) { return (path == PATH_X3 && e->rx_x3 && e->cls_kind == 1) ? 2 : 0; }

}  // namespace

extern "C" {

const char* dmad_last_error(void) { return g_err.c_str(); }
const char* dmad_version(void) { return "dmad-hip 0.5 (gfx950)"; }
const char* dmad_last_warning(void) { return g_warn.c_str(); }

int dmad_create(const dmad_config* cfg, dmad_engine** out) {
    if (!cfg || !out) return fail(DMAD_ERR_INVALID, "null argument");
    if (cfg->struct_size != (int32_t)sizeof(dmad_config))       // a caller built against another revision of dmad.h
        return fail(DMAD_ERR_INVALID, "dmad_config.struct_size is %d, this library's dmad_config has %d bytes (%s)", cfg->struct_size,
                    (int)sizeof(dmad_config), dmad_version());
    if (cfg->res_channels != 256 || cfg->skip_channels != 256)
        return fail(DMAD_ERR_INVALID, "only res_channels = skip_channels = 256 is supported (got %d/%d)", cfg->res_channels, cfg->skip_channels);
    if (cfg->embed_dim_in != 128 || cfg->embed_dim_mid != 512 || cfg->embed_dim_out != 512)
        return fail(DMAD_ERR_INVALID, "only step-embedding dims 128/512/512 are supported");
    if (cfg->num_res_layers < 1 || cfg->num_res_layers > 64) return fail(DMAD_ERR_INVALID, "num_res_layers %d outside [1,64]", cfg->num_res_layers);
    if (cfg->dilation_cycle < 1 || cfg->dilation_cycle > 12) return fail(DMAD_ERR_INVALID, "dilation_cycle %d outside [1,12]", cfg->dilation_cycle);
    if (cfg->clip_len < 128 || cfg->clip_len % 128) return fail(DMAD_ERR_INVALID, "clip_len %d must be a positive multiple of 128", cfg->clip_len);
    if (cfg->with_classifier && cfg->clip_len != 16000) return fail(DMAD_ERR_INVALID, "the mel front-end needs clip_len = 16000");
    if (cfg->max_batch < 1) return fail(DMAD_ERR_INVALID, "max_batch must be >= 1");
    if (cfg->precision != DMAD_BF16 && cfg->precision != DMAD_FP32 && cfg->precision != DMAD_EXACT)
        return fail(DMAD_ERR_INVALID, "unknown precision %d", cfg->precision);
    if (cfg->recheck_batch < 0) return fail(DMAD_ERR_INVALID, "recheck_batch %d < 0", cfg->recheck_batch);
    if (cfg->half_type != DMAD_HALF_BF16 && cfg->half_type != DMAD_HALF_F16) return fail(DMAD_ERR_INVALID, "unknown half_type %d", cfg->half_type);
    int ndev = 0;
    HIPCHK(hipGetDeviceCount(&ndev));
    if (ndev < 1) return fail(DMAD_ERR_HIP, "no HIP device visible");
    dmad_engine* e = new dmad_engine();
    e->cfg = *cfg;
    e->L = cfg->clip_len; e->LP = cfg->clip_len + 2 * kPad; e->NL = cfg->num_res_layers; e->maxB = cfg->max_batch;
    e->LPm = cfg->clip_len + 2048;
    e->bf16 = cfg->precision != DMAD_FP32;
    e->f32 = cfg->precision != DMAD_BF16;
    e->f16 = cfg->half_type == DMAD_HALF_F16;
    e->maxB32 = cfg->precision == DMAD_FP32 ? cfg->max_batch : (cfg->recheck_batch > 0 ? cfg->recheck_batch : 32);
    if (e->maxB32 > cfg->max_batch) e->maxB32 = cfg->max_batch;
    e->mode = cfg->precision == DMAD_EXACT ? DMAD_MODE_EXACT_VOTES : (cfg->precision == DMAD_FP32 ? DMAD_MODE_FP32 : DMAD_MODE_FAST);
    e->tau = cfg->half_type == DMAD_HALF_F16 ? 0.034f : 0.30f;  // measured logit-difference error (against the leader) of the 16-bit path x 1.4 (see dmad.h)
    e->tau2 = 1e-3f;                        // the same for the split-f16 tier (dmad_set_recheck_margin2)
    e->un_h16 = cfg->precision != DMAD_FP32;    // engines with a 16-bit side also get the UNet's f16 tier (once UNet weights are loaded)
    e->rx_h16 = cfg->precision != DMAD_FP32;    // ... and ResNeXt29's (once its weights are loaded)
    if (const char* v = getenv("DMAD_RX_H16")) if (v[0] == '0') e->rx_h16 = false;      // A/B switch: ResNeXt29 on the fp32 matrix cores in every tier
    e->rx_x3 = cfg->precision == DMAD_EXACT;    // ... and its split-f16 tier
    if (const char* v = getenv("DMAD_RX_X3")) if (v[0] == '0') e->rx_x3 = false;        // A/B switch
    e->tau_spec = 0.13f;                    // spec-domain vote loop: measured logit-difference error of the f16 UNet chain x headroom (see dmad.h)
    e->tau_spec2 = 5e-4f;                   // ... of the chain on the split-f16 tier (measured 2.3e-4)
    e->un_x3 = cfg->precision == DMAD_EXACT;    // exact-vote engines also hold the UNet's split-f16 middle tier
    const bool wn = cfg->with_wavenet != 0;
    if (e->bf16 && !wn_final_p_supported(cfg->num_res_layers)) {
        delete e;
        return fail(DMAD_ERR_INVALID, "bf16 path does not support num_res_layers = %d", cfg->num_res_layers);
    }
    const size_t B = e->maxB, L = e->L, LP = e->LP, NL = e->NL, B32 = e->maxB32;
    int r = 0;
    do {
        if ((r = gemm_f32_configure())) { r = fail(DMAD_ERR_HIP, "hipFuncSetAttribute(max dynamic LDS, fp32 narrow tile) failed: %d", r); break; }
        if ((r = e->alloc(&e->xt, B * L))) break;
        if ((r = e->alloc(&e->eps, B * L))) break;
        if ((r = e->alloc(&e->x0, B * L))) break;
        if ((r = e->alloc(&e->znoise, B * L))) break;
        if (e->bf16 && wn) {
            if ((r = e->alloc(&e->hA, B * LP * kC, true))) break;
            if ((r = e->alloc(&e->hB, B * LP * kC, true))) break;
            if ((r = e->alloc(&e->gstore, NL * B * L * kC))) break;
            if ((r = wn_bf16_configure())) { r = fail(DMAD_ERR_HIP, "hipFuncSetAttribute(max dynamic LDS) failed: %d", r); break; }
        }
        if (e->f32 && wn) {
            if ((r = e->alloc(&e->hA32, B32 * LP * kC, true))) break;
            if ((r = e->alloc(&e->hB32, B32 * LP * kC, true))) break;
            if ((r = e->alloc(&e->H32, B32 * L * 512))) break;
            if ((r = e->alloc(&e->g32, B32 * L * 256))) break;
            if ((r = e->alloc(&e->skip32, B32 * L * 256))) break;
            if (e->bf16 && (r = gemm_x3_configure())) { r = fail(DMAD_ERR_HIP, "hipFuncSetAttribute(max dynamic LDS, split-f16 tier) failed: %d", r); break; }
        }
        if (e->bf16 && e->f32) {             // recheck queue of the exact-vote mode
            e->rc_cap = 1l << 20;
            if (const char* q = getenv("DMAD_RECHECK_QUEUE")) {     // tests: a small queue exercises the mid-call drain
                const long v = atol(q);
                if (v >= 1 && v < e->rc_cap) e->rc_cap = v;
            }
            if (e->rc_cap < (long)e->maxB) e->rc_cap = e->maxB;    // one batch always fits: the drain condition needs no more
            if ((r = e->alloc(&e->rc_list, (size_t)e->rc_cap))) break;
            if ((r = e->alloc(&e->rc_list2, (size_t)e->rc_cap))) break;
            if ((r = e->alloc(&e->rc_n, 1, true))) break;
            hipError_t he = hipHostMalloc((void**)&e->rc_n_host, sizeof(unsigned long long), hipHostMallocDefault);
            if (he != hipSuccess) { r = fail(DMAD_ERR_HIP, "hipHostMalloc failed: %s", hipGetErrorString(he)); break; }
        }
        if (cfg->with_classifier) {
            if ((r = e->alloc(&e->mel_xp, B * e->LPm))) break;
            if ((r = e->alloc(&e->dftD, B * 32 * kDftLd))) break;
            if ((r = e->alloc(&e->melP, B * 32 * kMelLd))) break;
            if ((r = e->alloc(&e->melM, B * 32 * 32))) break;
            if ((r = e->alloc(&e->spec, B * 1024))) break;
            if ((r = e->alloc(&e->act0, B * 1024 * 64))) break;
            if ((r = e->alloc(&e->act1, B * 1024 * 64))) break;
            if ((r = e->alloc(&e->logits, B * cfg->num_classes))) break;
            e->slab_floats = 8l << 20;           // 32 MiB split-K workspace (largest user: 16 x [B*4][512])
            if (e->slab_floats < (long)B * 16 * 4096) e->slab_floats = (long)B * 16 * 4096;
            if ((r = e->alloc(&e->slab, (size_t)e->slab_floats))) break;
            if ((r = init_mel_constants(e))) break;
        }
    } while (0);
    if (r) { dmad_destroy(e); return r; }
    *out = e;
    return 0;
}

void dmad_destroy(dmad_engine* e) {
    if (!e) return;
    for (hipEvent_t ev : e->prof_ev) (void)hipEventDestroy(ev);
    for (hipEvent_t ev : e->prof_ev_f) (void)hipEventDestroy(ev);
    if (e->rc_n_host) (void)hipHostFree(e->rc_n_host);
    for (void* p : e->allocs) (void)hipFree(p);
    delete e;
}

int64_t dmad_device_bytes(const dmad_engine* e) { return e ? e->bytes : 0; }

int dmad_profile_layers(dmad_engine* e, int32_t max_launches) {
    if (!e || max_launches < 0) return fail(DMAD_ERR_INVALID, "bad argument");
    e->prof_used = e->prof_used_f = 0;
    e->prof_on = max_launches > 0;
    while (e->prof_ev.size() < (size_t)max_launches * 2) {
        hipEvent_t ev;
        HIPCHK(hipEventCreate(&ev));
        e->prof_ev.push_back(ev);
    }
    const size_t nf = ((size_t)max_launches + e->NL - 1) / (e->NL > 1 ? e->NL - 1 : 1) + 1;   // one final launch per NL-1 timed layer launches
    while (e->prof_ev_f.size() < nf * 2) {
        hipEvent_t ev;
        HIPCHK(hipEventCreate(&ev));
        e->prof_ev_f.push_back(ev);
    }
    return 0;
}

static int prof_sum(std::vector<hipEvent_t>& evs, size_t used, float* total_ms, int32_t* launches) {
    double tot = 0.0;
    const size_t pairs = used / 2;
    if (pairs) HIPCHK(hipEventSynchronize(evs[used - 1]));
    for (size_t i = 0; i < pairs; ++i) {
        float ms = 0.f;
        HIPCHK(hipEventElapsedTime(&ms, evs[2 * i], evs[2 * i + 1]));
        tot += ms;
    }
    *total_ms = (float)tot;
    *launches = (int32_t)pairs;
    return 0;
}

int dmad_profile_read(dmad_engine* e, float* total_ms, int32_t* launches) {
    if (!e || !total_ms || !launches) return fail(DMAD_ERR_INVALID, "null argument");
    CHK(prof_sum(e->prof_ev, e->prof_used, total_ms, launches));
    e->prof_used = 0;
    e->prof_on = false;
    return 0;
}

int dmad_profile_read_final(dmad_engine* e, float* total_ms, int32_t* launches) {
    if (!e || !total_ms || !launches) return fail(DMAD_ERR_INVALID, "null argument");
    CHK(prof_sum(e->prof_ev_f, e->prof_used_f, total_ms, launches));
    e->prof_used_f = 0;
    return 0;
}

int dmad_load_weight(dmad_engine* e, const char* name, const float* host, const int64_t* shape, int32_t ndim) {
    if (!e || !name || !host || !shape || ndim < 1 || ndim > 4) return fail(DMAD_ERR_INVALID, "bad argument to dmad_load_weight");
    int64_t n = 1;
    for (int i = 0; i < ndim; ++i) {
        if (shape[i] < 1) return fail(DMAD_ERR_INVALID, "weight '%s': non-positive dimension", name);
        n *= shape[i];
    }
    HostW& h = e->hw[name];
    h.v.assign(host, host + n);
    h.shape.assign(shape, shape + ndim);
    return 0;
}

int dmad_finalize_weights(dmad_engine* e) {
    if (!e) return fail(DMAD_ERR_INVALID, "null engine");
    // finalises whichever part (WaveNet, classifier) has its weights loaded and is not packed yet
    bool did = false;
    if (!e->wn_final && e->hw.count("init.w")) {
        if (!e->cfg.with_wavenet) return fail(DMAD_ERR_STATE, "engine was created with with_wavenet = 0: it has no WaveNet workspace");
        CHK(finalize_wavenet(e));
        e->wn_final = true; did = true;
    }
    if (e->cfg.with_classifier && !e->cls_final && e->hw.count("vgg.conv0.w")) {
        CHK(finalize_classifier(e));
        e->cls_final = true; e->cls_kind = 0; did = true;
    } else if (e->cfg.with_classifier && !e->cls_final && e->hw.count("rx.conv1.w")) {
        CHK(finalize_resnext(e));
        e->cls_final = true; e->cls_kind = 1; did = true;
    }
    if (!e->un_final && e->hw.count("un.time_embed.0.weight")) {
        if (!e->slab) {                     // engines created without a classifier have no split-K workspace yet
            e->slab_floats = 8l << 20;
            CHK(e->alloc(&e->slab, (size_t)e->slab_floats));
        }
        CHK(finalize_unet(e));
        e->un_final = true; did = true;
    }
    if (!did) return fail(DMAD_ERR_STATE, "nothing to finalise: no complete weight set was loaded");
    e->hw.clear();
    g_warn = e->warn;                       // dmad_last_warning(): empty unless this call found something to say
    e->warn.clear();
    return 0;
}

int dmad_wavenet_eps(dmad_engine* e, const float* x_t, int32_t t, int32_t B, float* eps, dmad_stream s) {
    if (!e || !x_t || !eps) return fail(DMAD_ERR_INVALID, "null argument");
    return wavenet_eps(e, x_t, t, B, eps, (hipStream_t)s, wave_path(e));
}

int dmad_one_shot(dmad_engine* e, const float* x_t, int32_t t, float c_a, float c_b, int32_t B, float* x0, dmad_stream s) {
    if (!e || !x_t || !x0) return fail(DMAD_ERR_INVALID, "null argument");
    CHK(wavenet_eps(e, x_t, t, B, e->eps, (hipStream_t)s, wave_path(e)));
    launch_lincomb(0, x_t, e->eps, nullptr, c_a, c_b, 0.f, x0, (long)B * e->L, (hipStream_t)s);
    LASTCHK();
    return 0;
}

int dmad_ddpm_step(dmad_engine* e, float* x, int32_t t, float c_eps, float c_div, float c_sig, const float* z, uint64_t seed,
                   uint64_t sample0, int32_t B, dmad_stream s) {
    if (!e || !x) return fail(DMAD_ERR_INVALID, "null argument");
    CHK(wavenet_eps(e, x, t, B, e->eps, (hipStream_t)s, wave_path(e)));
    const float* zz = nullptr;
    if (c_sig != 0.f) {
        zz = z;
        if (!zz) {
            launch_philox_normal(seed, sample0, 1u + (uint32_t)t, e->znoise, B, e->L, (hipStream_t)s);
            zz = e->znoise;
        }
    }
    launch_lincomb(2, x, e->eps, zz, c_eps, c_div, c_sig, x, (long)B * e->L, (hipStream_t)s);
    LASTCHK();
    return 0;
}

int dmad_diffuse(dmad_engine* e, const float* x0, float c_a, float c_b, const float* z, uint64_t seed, uint64_t sample0,
                 int32_t B, float* x_t, dmad_stream s) {
    if (!e || !x0 || !x_t) return fail(DMAD_ERR_INVALID, "null argument");
    if (B < 1 || B > e->maxB) return fail(DMAD_ERR_STATE, "batch %d outside [1, max_batch=%d]", B, e->maxB);
    const float* zz = z;
    if (!zz) {
        launch_philox_normal(seed, sample0, 0xD1FFu, e->znoise, B, e->L, (hipStream_t)s);
        zz = e->znoise;
    }
    launch_lincomb(1, x0, nullptr, zz, c_a, c_b, 0.f, x_t, (long)B * e->L, (hipStream_t)s);
    LASTCHK();
    return 0;
}

int dmad_unet_eps(dmad_engine* e, const float* x_t, int32_t t, int32_t B, float* eps, dmad_stream s) {
    if (!e || !x_t || !eps) return fail(DMAD_ERR_INVALID, "null argument");
    return unet_eps(e, x_t, t, B, eps, (hipStream_t)s);
}

int dmad_unet_eps_tier(dmad_engine* e, const float* x_t, int32_t t, int32_t B, int32_t tier, float* eps, dmad_stream s) {
    if (!e || !x_t || !eps) return fail(DMAD_ERR_INVALID, "null argument");
    if (tier != 0 && tier != 1 && tier != 2) return fail(DMAD_ERR_INVALID, "unknown UNet tier %d (0 exact fp32, 1 16-bit, 2 split-f16)", tier);
    return unet_eps(e, x_t, t, B, eps, (hipStream_t)s, tier);
}

int dmad_unet_p_sample(dmad_engine* e, float* x, int32_t t, float c_a, float c_b, float c_1, float c_2, float c_sig, const float* z,
                       uint64_t seed, uint64_t sample0, int32_t B, float* x0_out, dmad_stream s) {
    if (!e || !x) return fail(DMAD_ERR_INVALID, "null argument");
    CHK(unet_eps(e, x, t, B, e->un_eps, (hipStream_t)s));
    const float* zz = nullptr;
    if (c_sig != 0.f) {
        zz = z;
        if (!zz) {
            launch_philox_normal(seed, sample0, 0x0E70u + (uint32_t)t, e->znoise, B, 1024, (hipStream_t)s);
            zz = e->znoise;
        }
    }
    launch_unet_p_sample(x, e->un_eps, zz, c_a, c_b, c_1, c_2, c_sig, x, x0_out, (long)B * 1024, (hipStream_t)s);
    LASTCHK();
    return 0;
}

int dmad_ddpm_purify(dmad_engine* e, const float* x0, int32_t t_star, float c_a, float c_b, const float* c_eps, const float* c_div,
                     const float* c_sig, uint64_t seed, uint64_t sample0, int32_t B, float* out, dmad_stream s) {
    if (!e || !x0 || !out || !c_eps || !c_div || !c_sig) return fail(DMAD_ERR_INVALID, "null argument");
    if (t_star < 1) return fail(DMAD_ERR_INVALID, "t_star %d < 1", t_star);
    CHK(dmad_diffuse(e, x0, c_a, c_b, nullptr, seed, sample0, B, out, s));
    for (int t = t_star - 1; t >= 0; --t)
        CHK(dmad_ddpm_step(e, out, t, c_eps[t], c_div[t], t > 0 ? c_sig[t] : 0.f, nullptr, seed, sample0, B, s));
    return 0;
}

int dmad_mel_db(dmad_engine* e, const float* x, int32_t B, float* spec, dmad_stream s) {
    if (!e || !x || !spec) return fail(DMAD_ERR_INVALID, "null argument");
    return mel_db(e, x, B, spec, (hipStream_t)s);
}

int dmad_mel_power(dmad_engine* e, const float* x, int32_t B, float* mel, dmad_stream s) {
    if (!e || !x || !mel) return fail(DMAD_ERR_INVALID, "null argument");
    return mel_db(e, x, B, mel, (hipStream_t)s, 0);
}

int dmad_power_to_db(dmad_engine* e, const float* x, int64_t n, float* y, dmad_stream s) {
    if (!e || !x || !y || n < 0) return fail(DMAD_ERR_INVALID, "bad argument");
    if (n) launch_power_to_db(x, y, (long)n, (hipStream_t)s);
    LASTCHK();
    return 0;
}

int dmad_classify(dmad_engine* e, const float* spec, int32_t B, float* logits, dmad_stream s) {
    if (!e || !spec || !logits) return fail(DMAD_ERR_INVALID, "null argument");
    return classify(e, spec, B, logits, (hipStream_t)s);
}

int dmad_classify_tier(dmad_engine* e, const float* spec, int32_t B, int32_t tier, float* logits, dmad_stream s) {
    if (!e || !spec || !logits) return fail(DMAD_ERR_INVALID, "null argument");
    if (tier != 0 && tier != 1 && tier != 2) return fail(DMAD_ERR_INVALID, "unknown classifier tier %d (0 fp32, 1 16-bit, 2 split-f16)", tier);
    return classify(e, spec, B, logits, (hipStream_t)s, tier);
}

int dmad_conv_h16(const uint16_t* x, const uint16_t* x2, int32_t ksplit, const uint16_t* w, const float* bias, const uint16_t* res16,
                  int32_t B, int32_t H, int32_t M, int32_t K, int32_t taps, int32_t stride, int32_t groups, int32_t relu,
                  float* out32, uint16_t* out16, dmad_stream s) {
    if (!x || !w || (!out32 && !out16)) return fail(DMAD_ERR_INVALID, "null argument");
    if (B < 1 || H < 1 || M < 1 || K < 1 || groups < 1 || (stride != 1 && stride != 2)) return fail(DMAD_ERR_INVALID, "bad geometry");
    if (int r = gemm_h16_configure()) return fail(DMAD_ERR_HIP, "hipFuncSetAttribute(max dynamic LDS, f16 conv GEMM) failed: %d", r);
    const int Ho = (H - 1) / stride + 1;
    GemmH16Args g{};
    g.A = w; g.X = x; g.C = out32; g.C16 = out16; g.shift = bias; g.res16 = res16; g.M = M; g.K = K; g.taps = taps; g.ldc = groups * M;
    g.N = (long)B * Ho * Ho; g.H = H; g.W = H; g.ldx = x2 ? ksplit : groups * K; g.stride = stride; g.relu = relu; g.groups = groups;
    if (x2) { g.X2 = x2; g.ksplit = ksplit; g.ldx2 = K - ksplit; }
    launch_gemm_h16(g, (hipStream_t)s);
    LASTCHK();
    return 0;
}

int dmad_conv_h16_up2(const uint16_t* x_half, const uint16_t* w, const float* bias, const uint16_t* res16, int32_t B, int32_t H, int32_t M, int32_t K,
                      float* out32, uint16_t* out16, float* stats, dmad_stream s) {
    if (!x_half || !w || (!out32 && !out16)) return fail(DMAD_ERR_INVALID, "null argument");
    if (B < 1 || H < 2 || (H & 1) || M < 1 || K < 1) return fail(DMAD_ERR_INVALID, "bad geometry");
    if (int r = gemm_h16_configure()) return fail(DMAD_ERR_HIP, "hipFuncSetAttribute(max dynamic LDS, f16 conv GEMM) failed: %d", r);
    GemmH16Args g{};
    g.A = w; g.X = x_half; g.C = out32; g.C16 = out16; g.shift = bias; g.res16 = res16; g.M = M; g.K = K; g.taps = 9; g.ldc = M;
    g.N = (long)B * H * H; g.H = H; g.W = H; g.ldx = K; g.stride = 1; g.up2 = 1;
    if (stats) { g.stats = stats; g.stats_px = 64; }
    if (!gemm_h16_fuses_up2(g)) return fail(DMAD_ERR_STATE, "this shape is not served by the form that fuses the upsampling (the caller materialises the x2 map)");
    launch_gemm_h16(g, (hipStream_t)s);
    LASTCHK();
    return 0;
}

int dmad_split_f16(const float* x, int64_t n, float* y, dmad_stream s) {
    if (!x || !y || n < 0 || (n & 3)) return fail(DMAD_ERR_INVALID, "bad argument (n must be a multiple of 4)");
    if (n) launch_scale(x, 1.f, y, (long)n, (hipStream_t)s, true);
    LASTCHK();
    return 0;
}

int dmad_conv_x3(const float* x, const float* x2, int32_t ksplit, const float* w, const float* bias, const float* res, int32_t B, int32_t H,
                 int32_t M, int32_t K, int32_t taps, int32_t stride, int32_t groups, int32_t relu, int32_t out_split, int32_t res_split, float* out,
                 dmad_stream s) {
    if (!x || !w || !out) return fail(DMAD_ERR_INVALID, "null argument");
    if (B < 1 || H < 1 || M < 1 || K < 1 || groups < 1 || (stride != 1 && stride != 2) || (groups > 1 && x2)) return fail(DMAD_ERR_INVALID, "bad geometry");
    if (int r = gemm_x3_configure()) return fail(DMAD_ERR_HIP, "hipFuncSetAttribute(max dynamic LDS, split-f16 tier) failed: %d", r);
    const int Ho = (H - 1) / stride + 1;
    GemmF32Args g{};
    g.A = w; g.X = x; g.C = out; g.shift = bias; g.res = res; g.M = M; g.K = K; g.taps = taps; g.ldc = groups * M; g.relu = relu; g.N = (long)B * Ho * Ho;
    g.mode = 2; g.H = H; g.W = H; g.Cin = K; g.ldx = x2 ? ksplit : groups * K; g.stride = stride; g.x3 = 1; g.out_split = out_split; g.res_split = res_split;
    g.groups = groups;
    if (x2) { g.X2 = x2; g.ksplit = ksplit; g.ldx2 = K - ksplit; }
    launch_gemm_f32(g, (hipStream_t)s);
    LASTCHK();
    return 0;
}

int dmad_conv_h16_stats(const uint16_t* x, const uint16_t* w, const float* bias, const uint16_t* res16, int32_t B, int32_t H, int32_t M, int32_t K,
                        int32_t taps, int32_t stride, uint16_t* out16, float* stats, dmad_stream s) {
    if (!x || !w || !out16 || !stats) return fail(DMAD_ERR_INVALID, "null argument");
    if (B < 1 || H < 1 || M < 1 || K < 1 || (stride != 1 && stride != 2)) return fail(DMAD_ERR_INVALID, "bad geometry");
    if (int r = gemm_h16_configure()) return fail(DMAD_ERR_HIP, "hipFuncSetAttribute(max dynamic LDS, f16 conv GEMM) failed: %d", r);
    const int Ho = (H - 1) / stride + 1;
    if (Ho * Ho != 16 && (Ho * Ho) % 64) return fail(DMAD_ERR_INVALID, "statistics blocks need maps of 16 or a multiple of 64 pixels");
    GemmH16Args g{};
    g.A = w; g.X = x; g.C16 = out16; g.shift = bias; g.res16 = res16; g.M = M; g.K = K; g.taps = taps; g.ldc = M;
    g.N = (long)B * Ho * Ho; g.H = H; g.W = H; g.ldx = K; g.stride = stride; g.stats = stats; g.stats_px = Ho * Ho >= 64 ? 64 : 16;
    launch_gemm_h16(g, (hipStream_t)s);
    LASTCHK();
    return 0;
}

int dmad_groupnorm16_apply(const uint16_t* x, const float* st, const uint16_t* x2, const float* st2, int32_t c1, const float* gamma,
                           const float* beta, const float* ss, int32_t silu, int32_t B, int32_t HW, int32_t C, uint16_t* y16, float* y32,
                           dmad_stream s) {
    if (!x || !st || !gamma || !beta || (!y16 && !y32)) return fail(DMAD_ERR_INVALID, "null argument");
    if (launch_groupnorm16_apply(x, st, x2, st2, c1, gamma, beta, ss, silu, y16, y32, B, HW, C, (hipStream_t)s))
        return fail(DMAD_ERR_INVALID, "no one-pass GroupNorm for a %d-pixel x %d-channel map", HW, C);
    LASTCHK();
    return 0;
}

int dmad_vote(dmad_engine* e, const float* logits, int32_t B, int64_t* counts, dmad_stream s) {
    if (!e || !logits || !counts || B < 1) return fail(DMAD_ERR_INVALID, "bad argument to dmad_vote");
    launch_vote(logits, B, e->cfg.num_classes, (unsigned long long*)counts, nullptr, (hipStream_t)s);
    LASTCHK();
    return 0;
}

int dmad_set_mode(dmad_engine* e, int32_t mode) {
    if (!e) return fail(DMAD_ERR_INVALID, "null engine");
    if (!(e->bf16 && e->f32)) {
        const int only = e->bf16 ? DMAD_MODE_FAST : DMAD_MODE_FP32;
        if (mode == only) return 0;
        return fail(DMAD_ERR_STATE, "mode %d needs a DMAD_EXACT engine (this one has only its %s path)", mode, e->bf16 ? "bf16" : "fp32");
    }
    if (mode != DMAD_MODE_FAST && mode != DMAD_MODE_EXACT_VOTES && mode != DMAD_MODE_FP32) return fail(DMAD_ERR_INVALID, "unknown mode %d", mode);
    e->mode = mode;
    return 0;
}

int dmad_set_waveform_tier(dmad_engine* e, int32_t tier) {
    if (!e) return fail(DMAD_ERR_INVALID, "null engine");
    if (tier != PATH_DEFAULT && tier != PATH_FP32 && tier != PATH_X3) return fail(DMAD_ERR_INVALID, "unknown waveform tier %d (0 16-bit, 1 fp32, 2 split-f16)", tier);
    if (!(e->bf16 && e->f32)) {
        if (tier == PATH_DEFAULT) return 0;
        return fail(DMAD_ERR_STATE, "waveform tiers need a DMAD_EXACT engine (this one has only its %s path)", e->bf16 ? "16-bit" : "fp32");
    }
    e->wave_tier = tier;
    return 0;
}

int dmad_set_recheck_margin(dmad_engine* e, float tau) {
    if (!e) return fail(DMAD_ERR_INVALID, "null engine");
    if (!(tau >= 0.f)) return fail(DMAD_ERR_INVALID, "recheck margin must be >= 0");
    e->tau = tau;
    return 0;
}

int dmad_set_recheck_margin2(dmad_engine* e, float tau2) {
    if (!e) return fail(DMAD_ERR_INVALID, "null engine");
    if (tau2 != tau2) return fail(DMAD_ERR_INVALID, "recheck margin is NaN");
    e->tau2 = tau2;                          // < 0: no middle tier, the queued samples go straight to the fp32 path
    return 0;
}

int dmad_recheck_stats(dmad_engine* e, int64_t* samples, int64_t* rechecked, int64_t* rechecked_fp32, int32_t reset) {
    if (!e) return fail(DMAD_ERR_INVALID, "null engine");
    if (samples) *samples = e->st_samples;
    if (rechecked) *rechecked = e->st_rechecked;
    if (rechecked_fp32) *rechecked_fp32 = e->st_rechecked2;
    if (reset) e->st_samples = e->st_rechecked = e->st_rechecked2 = 0;
    return 0;
}

int dmad_wavenet_eps_path(dmad_engine* e, const float* x_t, int32_t t, int32_t B, int32_t path, float* eps, dmad_stream s) {
    if (!e || !x_t || !eps) return fail(DMAD_ERR_INVALID, "null argument");
    if (path != PATH_DEFAULT && path != PATH_FP32 && path != PATH_X3) return fail(DMAD_ERR_INVALID, "unknown path %d", path);
    if (path != PATH_DEFAULT && !(e->bf16 && e->f32)) return fail(DMAD_ERR_STATE, "explicit WaveNet paths need a DMAD_EXACT engine");
    return wavenet_eps(e, x_t, t, B, eps, (hipStream_t)s, path);
}

int dmad_debug_rounding(dmad_engine* e, const int32_t masks[5]) {
    if (!e || !masks) return fail(DMAD_ERR_INVALID, "null argument");
    if (!(e->bf16 && e->f32)) return fail(DMAD_ERR_STATE, "dmad_debug_rounding needs a DMAD_EXACT engine (it acts on the split-f16 tier)");
    for (int i = 0; i < 5; ++i) {
        if (masks[i] < 0 || masks[i] > 7) return fail(DMAD_ERR_INVALID, "mask %d = %d outside [0, 7]", i, masks[i]);
        e->diag[i] = masks[i];
    }
    return 0;
}

int dmad_eval_samples(dmad_engine* e, const float* clip, float sigma, float sqrt_alpha_bar_star, int32_t t, float c_a, float c_b,
                      uint64_t seed, uint64_t sample0, const float* delta, const int64_t* idx, int64_t n, int32_t path, float* logits_out,
                      float* x0_out, dmad_stream s) {
    if (!e || !clip || !idx || (!logits_out && !x0_out)) return fail(DMAD_ERR_INVALID, "null argument");
    if (n < 0) return fail(DMAD_ERR_INVALID, "n < 0");
    if (path != PATH_DEFAULT && path != PATH_FP32 && path != PATH_X3) return fail(DMAD_ERR_INVALID, "unknown path %d", path);
    if (path != PATH_DEFAULT && !(e->bf16 && e->f32)) return fail(DMAD_ERR_STATE, "explicit WaveNet paths need a DMAD_EXACT engine");
    if (logits_out && !e->cfg.with_classifier) return fail(DMAD_ERR_STATE, "engine was created with with_classifier = 0");
    hipStream_t st = (hipStream_t)s;
    const int L = e->L, C = e->cfg.num_classes;
    const int cap = path == PATH_DEFAULT ? e->maxB : e->maxB32;
    for (int64_t done = 0; done < n; done += cap) {
        const int B = (int)(n - done < cap ? n - done : cap);
        launch_mc_noise_scale_idx(clip, delta, sigma, sqrt_alpha_bar_star, seed, sample0, (const long long*)idx + done, e->xt, B, L, st);
        CHK(wavenet_eps(e, e->xt, t, B, e->eps, st, path));
        float* x0 = x0_out ? x0_out + done * L : e->x0;
        launch_lincomb(0, e->xt, e->eps, nullptr, c_a, c_b, 0.f, x0, (long)B * L, st);
        if (logits_out) {
            CHK(mel_db(e, x0, B, e->spec, st));
            CHK(classify(e, e->spec, B, logits_out + done * C, st, path == PATH_DEFAULT ? cls_tier(e) : cls_tier_of_path(e, path)));
        }
    }
    LASTCHK();
    return 0;
}

}  // extern "C"

namespace {

// The queued samples of an exact-vote pass, re-evaluated on the exact-fp32 WaveNet from the same noise.  Waits for the
// stream once (the queue length decides the launches).
struct RecheckJob {
    const float* clip; const float* delta; float sigma, scale; int t; float c_a, c_b; uint64_t seed, sample0;
    int64_t* counts; float* logits_out; float* x0_out;
};
// one tier of the recheck: rows idx[0..n) re-evaluated on `path`; tau >= 0: rows whose margin is still below tau are queued in
// `next` (they do not vote), tau < 0: every row votes
int recheck_pass(dmad_engine* e, const RecheckJob& j, const long long* list, long n, int path, float tau, long long* next, hipStream_t st) {
    const int L = e->L, C = e->cfg.num_classes;
    for (long done = 0; done < n; done += e->maxB32) {
        const int B = (int)(n - done < e->maxB32 ? n - done : e->maxB32);
        const long long* idx = list + done;
        launch_mc_noise_scale_idx(j.clip, j.delta, j.sigma, j.scale, j.seed, j.sample0, idx, e->xt, B, L, st);
        CHK(wavenet_eps(e, e->xt, j.t, B, e->eps, st, path));
        launch_lincomb(0, e->xt, e->eps, nullptr, j.c_a, j.c_b, 0.f, e->x0, (long)B * L, st);
        if (j.x0_out) launch_scatter_rows(e->x0, idx, (long long)j.sample0, j.x0_out, B, L, st);
        CHK(mel_db(e, e->x0, B, e->spec, st));
        CHK(classify(e, e->spec, B, e->logits, st, cls_tier_of_path(e, path)));
        if (j.logits_out) launch_scatter_rows(e->logits, idx, (long long)j.sample0, j.logits_out, B, C, st);
        if (tau >= 0.f) launch_vote_margin(e->logits, B, C, (unsigned long long*)j.counts, tau, 0, idx, next, e->rc_n, e->rc_cap, nullptr, st);
        else launch_vote(e->logits, B, C, (unsigned long long*)j.counts, nullptr, st);
    }
    LASTCHK();
    return 0;
}

long read_queue_length(dmad_engine* e, hipStream_t st, int* rc) {
    *rc = 0;
    if (hipMemcpyAsync(e->rc_n_host, e->rc_n, sizeof(unsigned long long), hipMemcpyDeviceToHost, st) != hipSuccess ||
        hipMemsetAsync(e->rc_n, 0, sizeof(unsigned long long), st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) {
        *rc = fail(DMAD_ERR_HIP, "reading the recheck queue length failed");
        return 0;
    }
    const long n = (long)*e->rc_n_host;
    if (n > e->rc_cap) *rc = fail(DMAD_ERR_STATE, "recheck queue overflow (%ld > %ld)", n, e->rc_cap);
    return n;
}

// The samples the 16-bit pass queued: tier 2 = the fp32 pipeline on split-f16 operands (three MFMAs per product, ~fp32
// accuracy at several times the fp32 matrix rate) settles every sample whose margin exceeds ITS error bound tau2; what is
// left (margins inside tau2) goes to tier 3, the exact-fp32 path.  Two stream synchronisations (the queue lengths decide
// the launches).
int run_recheck(dmad_engine* e, const RecheckJob& j, hipStream_t st) {
    int rc = 0;
    const long n1 = read_queue_length(e, st, &rc);
    if (rc) return rc;
    e->st_rechecked += n1;
    if (n1 == 0) return 0;
    if (e->tau2 >= 0.f && e->wdil_x3) {
        CHK(recheck_pass(e, j, e->rc_list, n1, PATH_X3, e->tau2, e->rc_list2, st));
        const long n2 = read_queue_length(e, st, &rc);
        if (rc) return rc;
        e->st_rechecked2 += n2;
        if (n2) CHK(recheck_pass(e, j, e->rc_list2, n2, PATH_FP32, -1.f, nullptr, st));
    } else {
        e->st_rechecked2 += n1;
        CHK(recheck_pass(e, j, e->rc_list, n1, PATH_FP32, -1.f, nullptr, st));
    }
    return 0;
}

}  // namespace

extern "C" {

int dmad_smooth_votes(dmad_engine* e, const float* clip, float sigma, float sqrt_alpha_bar_star, int32_t t, float c_a,
                      float c_b, int64_t n, int32_t batch, uint64_t seed, uint64_t sample0, const float* delta, int64_t* counts,
                      float* logits_out, float* x0_out, dmad_stream s) {
    if (!e || !clip) return fail(DMAD_ERR_INVALID, "null argument");
    if (n < 0 || batch < 1 || batch > e->maxB) return fail(DMAD_ERR_STATE, "batch %d outside [1, max_batch=%d] or n < 0", batch, e->maxB);
    if (e->cfg.with_classifier && !counts) return fail(DMAD_ERR_INVALID, "counts must not be null");
    hipStream_t st = (hipStream_t)s;
    const int L = e->L, C = e->cfg.num_classes;
    const bool recheck = e->bf16 && e->f32 && e->mode == DMAD_MODE_EXACT_VOTES && e->cfg.with_classifier;
    int64_t queued_from = 0;               // first sample (relative) of the current recheck segment
    // an earlier call that failed between queueing and draining must not leave its indices to this one
    if (recheck) HIPCHK(hipMemsetAsync(e->rc_n, 0, sizeof(unsigned long long), st));
    for (int64_t done = 0; done < n; done += batch) {
        const int B = (int)((n - done < batch) ? (n - done) : batch);
        launch_mc_noise_scale(clip, delta ? delta + done * L : nullptr, sigma, sqrt_alpha_bar_star, seed, sample0 + (uint64_t)done,
                              e->xt, B, L, st);
        CHK(wavenet_eps(e, e->xt, t, B, e->eps, st));
        float* x0 = x0_out ? x0_out + done * L : e->x0;
        launch_lincomb(0, e->xt, e->eps, nullptr, c_a, c_b, 0.f, x0, (long)B * L, st);
        if (e->cfg.with_classifier) {
            CHK(mel_db(e, x0, B, e->spec, st));
            float* lg = logits_out ? logits_out + done * C : e->logits;
            CHK(classify(e, e->spec, B, lg, st, cls_tier(e)));
            if (recheck) {
                launch_vote_margin(lg, B, C, (unsigned long long*)counts, e->tau, (long long)(sample0 + (uint64_t)done), nullptr, e->rc_list,
                                   e->rc_n, e->rc_cap, nullptr, st);
                // the queue holds at most rc_cap indices: drain it before the samples voted since the last drain could overflow it
                if (done + B - queued_from + batch > e->rc_cap && done + B < n) {
                    CHK(run_recheck(e, RecheckJob{clip, delta, sigma, sqrt_alpha_bar_star, t, c_a, c_b, seed, sample0, counts, logits_out, x0_out}, st));
                    queued_from = done + B;
                }
            } else {
                launch_vote(lg, B, C, (unsigned long long*)counts, nullptr, st);
            }
        }
    }
    if (recheck && n > 0)
        CHK(run_recheck(e, RecheckJob{clip, delta, sigma, sqrt_alpha_bar_star, t, c_a, c_b, seed, sample0, counts, logits_out, x0_out}, st));
    if (e->cfg.with_classifier) e->st_samples += n;
    LASTCHK();
    return 0;
}

}  // extern "C"

namespace {

struct SpecJob {
    const float* clip; float sigma; int t_star; float q_a, q_b; const float *c_a, *c_b, *c_1, *c_2, *c_sig; float mel_lo, mel_hi;
    uint64_t seed;
};
// One batch of the spec-domain chain (include/dmad.h, dmad_spec_smooth_votes): rows are samples s0 + b, or idx[b] when an index
// list is given (the recheck pass).  h16: -1 the mode's UNet tier, 0 exact fp32, 1 the 16-bit tier.  The purified dB spectrograms
// land in sp, the logits in lg.
// the chain behind its noisy waveforms: e->xt [B][L] -> purified dB spectrograms sp, logits lg.  cls16: the classifier tier (vote loops' first
// pass: the 16-bit tier where one is resident; every other caller: fp32)
int spec_chain_from_xt(dmad_engine* e, const SpecJob& j, uint64_t s0, const long long* idx, int B, int h16, int cls16, float* sp, float* lg, hipStream_t st) {
    CHK(mel_db(e, e->xt, B, e->spec, st));
    launch_philox_normal(j.seed, s0, 0x5BECu, e->znoise, B, 1024, st, idx);
    float* x = e->x0;                                                                   // [B][32][32] chain state
    launch_spec_diffuse(e->spec, e->znoise, j.mel_lo, j.mel_hi, j.q_a, j.q_b, x, (long)B * 1024, st);
    for (int t = j.t_star; t >= 0; --t) {
        CHK(unet_eps(e, x, t, B, e->un_eps, st, h16));
        const float sig = t > 0 ? j.c_sig[t] : 0.f;
        if (sig != 0.f) launch_philox_normal(j.seed, s0, 0x0E70u + (uint32_t)t, e->znoise, B, 1024, st, idx);
        launch_unet_p_sample(x, e->un_eps, sig != 0.f ? e->znoise : nullptr, j.c_a[t], j.c_b[t], j.c_1[t], j.c_2[t], sig, x, nullptr, (long)B * 1024, st);
    }
    launch_spec_unstandardize(x, j.mel_lo, j.mel_hi, sp, (long)B * 1024, st);
    CHK(classify(e, sp, B, lg, st, cls16));
    return 0;
}

int spec_chain(dmad_engine* e, const SpecJob& j, uint64_t s0, const long long* idx, int B, int h16, float* sp, float* lg, hipStream_t st) {
    const int L = e->L;
    if (idx) launch_mc_noise_scale_idx(j.clip, nullptr, j.sigma, 1.f, j.seed, 0, idx, e->xt, B, L, st);
    else launch_mc_noise_scale(j.clip, nullptr, j.sigma, 1.f, j.seed, s0, e->xt, B, L, st);      // no wave denoiser: no sqrt(alpha_bar*) scale
    return spec_chain_from_xt(e, j, s0, idx, B, h16, h16 == 1 ? cls_tier(e) : 0, sp, lg, st);      // the recheck tiers: the fp32 classifier
}

}  // namespace

extern "C" {

int dmad_spec_smooth_votes(dmad_engine* e, const float* clip, float sigma, int32_t t_star, float q_a, float q_b, const float* c_a,
                           const float* c_b, const float* c_1, const float* c_2, const float* c_sig, float mel_lo, float mel_hi, int64_t n,
                           int32_t batch, uint64_t seed, uint64_t sample0, int64_t* counts, float* logits_out, float* spec_out, dmad_stream s) {
    if (!e || !clip || !counts || !c_a || !c_b || !c_1 || !c_2 || !c_sig) return fail(DMAD_ERR_INVALID, "null argument");
    if (!e->cfg.with_classifier || !e->cls_final) return fail(DMAD_ERR_STATE, "the spec-domain vote loop needs the mel front-end and a finalised classifier");
    if (!e->un_final) return fail(DMAD_ERR_STATE, "UNet weights are not finalised (dmad_load_weight + dmad_finalize_weights)");
    if (n < 0 || batch < 1 || batch > e->maxB) return fail(DMAD_ERR_STATE, "batch %d outside [1, max_batch=%d] or n < 0", batch, e->maxB);
    if (t_star < 0) return fail(DMAD_ERR_INVALID, "t_star %d < 0", t_star);
    if (!(mel_hi > mel_lo)) return fail(DMAD_ERR_INVALID, "empty mel range");
    hipStream_t st = (hipStream_t)s;
    const int C = e->cfg.num_classes;
    const SpecJob job{clip, sigma, t_star, q_a, q_b, c_a, c_b, c_1, c_2, c_sig, mel_lo, mel_hi, seed};
    // exact-vote mode of a DMAD_EXACT engine: the chain runs on the UNet's 16-bit tier, a sample whose top-2 margin is below
    // tau_spec is queued and its WHOLE chain is re-run on the exact-fp32 UNet from the same Philox keys
    const bool recheck = e->bf16 && e->f32 && e->un_h16 && e->mode == DMAD_MODE_EXACT_VOTES;
    if (recheck) HIPCHK(hipMemsetAsync(e->rc_n, 0, sizeof(unsigned long long), st));
    // The queued samples: tier 2 = the whole chain again on the UNet's split-f16 tier (fp32-grade at several times the fp32 matrix rate)
    // settles every sample whose margin exceeds ITS bound tau_spec2; what is left goes to tier 3, the exact-fp32 UNet.  Every pass runs
    // its samples in batches of up to max_batch from the same Philox keys; a re-evaluated row of logits_out / spec_out carries the last
    // tier's result.
    auto pass = [&](const long long* list, long nq, int tier, float tau, long long* next) -> int {
        for (long done = 0; done < nq; done += e->maxB) {
            const int B = (int)(nq - done < e->maxB ? nq - done : e->maxB);
            const long long* idx = list + done;
            CHK(spec_chain(e, job, 0, idx, B, tier, e->spec, e->logits, st));
            if (spec_out) launch_scatter_rows(e->spec, idx, (long long)sample0, spec_out, B, 1024, st);
            if (logits_out) launch_scatter_rows(e->logits, idx, (long long)sample0, logits_out, B, C, st);
            if (tau >= 0.f) launch_vote_margin(e->logits, B, C, (unsigned long long*)counts, tau, 0, idx, next, e->rc_n, e->rc_cap, nullptr, st);
            else launch_vote(e->logits, B, C, (unsigned long long*)counts, nullptr, st);
        }
        return 0;
    };
    auto drain = [&]() -> int {
        int rc = 0;
        const long nq = read_queue_length(e, st, &rc);
        if (rc) return rc;
        e->st_spec_rechecked += nq;
        if (nq == 0) return 0;
        if (e->un_x3 && e->tau_spec2 >= 0.f) {
            CHK(pass(e->rc_list, nq, 2, e->tau_spec2, e->rc_list2));
            const long n2 = read_queue_length(e, st, &rc);
            if (rc) return rc;
            e->st_spec_rechecked2 += n2;
            if (n2) CHK(pass(e->rc_list2, n2, 0, -1.f, nullptr));
        } else {
            e->st_spec_rechecked2 += nq;
            CHK(pass(e->rc_list, nq, 0, -1.f, nullptr));
        }
        return 0;
    };
    int64_t queued_from = 0;
    for (int64_t done = 0; done < n; done += batch) {
        const int B = (int)((n - done < batch) ? (n - done) : batch);
        const uint64_t s0 = sample0 + (uint64_t)done;
        float* sp = spec_out ? spec_out + done * 1024 : e->spec;
        float* lg = logits_out ? logits_out + done * C : e->logits;
        CHK(spec_chain(e, job, s0, nullptr, B, (e->un_h16 && e->mode != DMAD_MODE_FP32) ? 1 : 0, sp, lg, st));     // first pass: the 16-bit tier
        if (recheck) {
            launch_vote_margin(lg, B, C, (unsigned long long*)counts, e->tau_spec, (long long)s0, nullptr, e->rc_list, e->rc_n, e->rc_cap, nullptr, st);
            if (done + B - queued_from + batch > e->rc_cap && done + B < n) { CHK(drain()); queued_from = done + B; }
        } else {
            launch_vote(lg, B, C, (unsigned long long*)counts, nullptr, st);
        }
    }
    if (recheck && n > 0) CHK(drain());
    e->st_spec_samples += n;
    LASTCHK();
    return 0;
}

int dmad_spec_eval_samples(dmad_engine* e, const float* clip, float sigma, int32_t t_star, float q_a, float q_b, const float* c_a,
                           const float* c_b, const float* c_1, const float* c_2, const float* c_sig, float mel_lo, float mel_hi, uint64_t seed,
                           const int64_t* idx, int64_t n, int32_t tier, float* logits_out, float* spec_out, dmad_stream s) {
    if (!e || !clip || !idx || !c_a || !c_b || !c_1 || !c_2 || !c_sig || (!logits_out && !spec_out)) return fail(DMAD_ERR_INVALID, "null argument");
    if (!e->cfg.with_classifier || !e->cls_final) return fail(DMAD_ERR_STATE, "the spec-domain chain needs the mel front-end and a finalised classifier");
    if (!e->un_final) return fail(DMAD_ERR_STATE, "UNet weights are not finalised (dmad_load_weight + dmad_finalize_weights)");
    if (n < 0 || t_star < 0 || !(mel_hi > mel_lo)) return fail(DMAD_ERR_INVALID, "bad argument");
    if (tier != 0 && tier != 1 && tier != 2) return fail(DMAD_ERR_INVALID, "unknown UNet tier %d (0 exact fp32, 1 16-bit, 2 split-f16)", tier);
    if (tier == 1 && !e->un_h16) return fail(DMAD_ERR_STATE, "this engine has no 16-bit UNet tier (DMAD_FP32 precision)");
    if (tier == 2 && !e->un_x3) return fail(DMAD_ERR_STATE, "this engine has no split-f16 UNet tier (it needs DMAD_EXACT precision)");
    hipStream_t st = (hipStream_t)s;
    const int C = e->cfg.num_classes;
    const SpecJob job{clip, sigma, t_star, q_a, q_b, c_a, c_b, c_1, c_2, c_sig, mel_lo, mel_hi, seed};
    for (int64_t done = 0; done < n; done += e->maxB) {
        const int B = (int)(n - done < e->maxB ? n - done : e->maxB);
        float* sp = spec_out ? spec_out + done * 1024 : e->spec;
        float* lg = logits_out ? logits_out + done * C : e->logits;
        CHK(spec_chain(e, job, 0, (const long long*)idx + done, B, tier, sp, lg, st));
    }
    LASTCHK();
    return 0;
}

int dmad_set_spec_recheck_margin(dmad_engine* e, float tau) {
    if (!e) return fail(DMAD_ERR_INVALID, "null engine");
    if (!(tau >= 0.f)) return fail(DMAD_ERR_INVALID, "recheck margin must be >= 0");
    e->tau_spec = tau;
    return 0;
}

int dmad_spec_recheck_stats(dmad_engine* e, int64_t* samples, int64_t* rechecked, int32_t reset) {
    if (!e) return fail(DMAD_ERR_INVALID, "null engine");
    if (samples) *samples = e->st_spec_samples;
    if (rechecked) *rechecked = e->st_spec_rechecked;
    if (reset) e->st_spec_samples = e->st_spec_rechecked = e->st_spec_rechecked2 = 0;
    return 0;
}

int dmad_set_spec_recheck_margin2(dmad_engine* e, float tau2) {
    if (!e) return fail(DMAD_ERR_INVALID, "null engine");
    if (tau2 != tau2) return fail(DMAD_ERR_INVALID, "recheck margin is NaN");
    e->tau_spec2 = tau2;                     // < 0: no middle tier, the queued samples go straight to the exact-fp32 UNet
    return 0;
}

int dmad_spec_recheck_stats2(dmad_engine* e, int64_t* samples, int64_t* rechecked, int64_t* rechecked_fp32, int32_t reset) {
    if (!e) return fail(DMAD_ERR_INVALID, "null engine");
    if (rechecked_fp32) *rechecked_fp32 = e->st_spec_rechecked2;
    return dmad_spec_recheck_stats(e, samples, rechecked, reset);
}

int dmad_query_logits(dmad_engine* e, const float* x, int32_t B, int32_t repeats, int32_t sampler, int32_t t_star, float c_a, float c_b,
                      const float* c_eps, const float* c_div, const float* c_sig, uint64_t seed, uint64_t sample0, float* logits,
                      int32_t* decisions, dmad_stream s) {
    if (!e || !x || !logits) return fail(DMAD_ERR_INVALID, "null argument");
    if (B < 1 || repeats < 1) return fail(DMAD_ERR_INVALID, "B and repeats must be >= 1");
    if (sampler < 0 || sampler > 2) return fail(DMAD_ERR_INVALID, "unknown sampler %d (0 none, 1 DDPM, 2 one-shot)", sampler);
    if (sampler && t_star < 1) return fail(DMAD_ERR_INVALID, "t_star %d < 1", t_star);
    if (sampler == 1 && (!c_eps || !c_div || !c_sig)) return fail(DMAD_ERR_INVALID, "the DDPM sampler needs its coefficient arrays");
    if (!e->cfg.with_classifier) return fail(DMAD_ERR_STATE, "engine was created with with_classifier = 0");
    hipStream_t st = (hipStream_t)s;
    const int L = e->L, C = e->cfg.num_classes;
    const long rows = (long)B * repeats;
    for (long r0 = 0; r0 < rows; r0 += e->maxB) {
        const int nb = (int)(rows - r0 < e->maxB ? rows - r0 : e->maxB);
        launch_repeat_rows(x, e->xt, B, r0, nb, L, st);
        const float* pur = e->xt;
        if (sampler == 1) {
            CHK(dmad_ddpm_purify(e, e->xt, t_star, c_a, c_b, c_eps, c_div, c_sig, seed, sample0 + (uint64_t)r0, nb, e->x0, s));
            pur = e->x0;
        } else if (sampler == 2) {
            CHK(wavenet_eps(e, e->xt, t_star - 1, nb, e->eps, st, wave_path(e)));
            launch_lincomb(0, e->xt, e->eps, nullptr, c_a, c_b, 0.f, e->x0, (long)nb * L, st);
            pur = e->x0;
        }
        CHK(mel_db(e, pur, nb, e->spec, st));
        CHK(classify(e, e->spec, nb, logits + r0 * C, st));       // the fp32 classifier, like AcousticSystem.forward's own call
        if (decisions) launch_vote(logits + r0 * C, nb, C, nullptr, decisions + r0, st);
    }
    LASTCHK();
    return 0;
}

int dmad_spec_query_logits(dmad_engine* e, const float* x, int32_t B, int32_t repeats, int32_t t_star, float q_a, float q_b, const float* c_a,
                           const float* c_b, const float* c_1, const float* c_2, const float* c_sig, float mel_lo, float mel_hi, uint64_t seed,
                           uint64_t sample0, float* logits, int32_t* decisions, dmad_stream s) {
    if (!e || !x || !logits || !c_a || !c_b || !c_1 || !c_2 || !c_sig) return fail(DMAD_ERR_INVALID, "null argument");
    if (B < 1 || repeats < 1) return fail(DMAD_ERR_INVALID, "B and repeats must be >= 1");
    if (!e->cfg.with_classifier || !e->cls_final) return fail(DMAD_ERR_STATE, "the spec-domain query needs the mel front-end and a finalised classifier");
    if (!e->un_final) return fail(DMAD_ERR_STATE, "UNet weights are not finalised (dmad_load_weight + dmad_finalize_weights)");
    if (t_star < 0 || !(mel_hi > mel_lo)) return fail(DMAD_ERR_INVALID, "bad argument");
    hipStream_t st = (hipStream_t)s;
    const int L = e->L, C = e->cfg.num_classes;
    const SpecJob job{nullptr, 0.f, t_star, q_a, q_b, c_a, c_b, c_1, c_2, c_sig, mel_lo, mel_hi, seed};
    const long rows = (long)B * repeats;
    for (long r0 = 0; r0 < rows; r0 += e->maxB) {
        const int nb = (int)(rows - r0 < e->maxB ? rows - r0 : e->maxB);
        launch_repeat_rows(x, e->xt, B, r0, nb, L, st);
        // the UNet tier of the map-returning surfaces (exact fp32 unless the engine is in its fast mode), the fp32 classifier: like
        // dmad_query_logits, a query hands logits back and has no recheck
        CHK(spec_chain_from_xt(e, job, sample0 + (uint64_t)r0, nullptr, nb, -1, 0, e->spec, logits + r0 * C, st));
        if (decisions) launch_vote(logits + r0 * C, nb, C, nullptr, decisions + r0, st);
    }
    LASTCHK();
    return 0;
}

int dmad_philox_raw(dmad_engine* e, uint64_t seed, uint64_t sample, uint32_t stream, uint32_t nblocks, uint32_t* out, dmad_stream s) {
    if (!e || !out) return fail(DMAD_ERR_INVALID, "null argument");
    launch_philox_raw(seed, sample, stream, nblocks, out, (hipStream_t)s);
    LASTCHK();
    return 0;
}

int dmad_philox_normal(dmad_engine* e, uint64_t seed, uint64_t sample0, uint32_t stream, int32_t B, float* z, dmad_stream s) {
    if (!e || !z || B < 1) return fail(DMAD_ERR_INVALID, "bad argument");
    launch_philox_normal(seed, sample0, stream, z, B, e->L, (hipStream_t)s);
    LASTCHK();
    return 0;
}

int dmad_time_layer(dmad_engine* e, int32_t layer, int32_t B, int32_t iters, float* ms_per_launch, dmad_stream s) {
    if (!e || !ms_per_launch || iters < 1) return fail(DMAD_ERR_INVALID, "bad argument");
    if (!e->wn_final || !e->bf16) return fail(DMAD_ERR_STATE, "dmad_time_layer needs a finalised bf16 engine");
    if (B < 1 || B > e->maxB || layer < 0 || layer >= e->NL) return fail(DMAD_ERR_STATE, "bad batch or layer");
    hipStream_t st = (hipStream_t)s;
    WnLayerArgs a{};
    a.hin = e->hA; a.hout = e->hB; a.gout = e->gstore + (size_t)layer * B * e->L * kC;
    a.w1p = e->w1p + (size_t)layer * 24 * 512 * 32; a.w2p = e->w2p + (size_t)layer * 8 * 256 * 32;
    a.b1 = e->b1p + (size_t)layer * 512; a.epi_c = e->epi_c + (size_t)layer * 256;
    a.dilation = 1 << (layer % e->cfg.dilation_cycle); a.L = e->L; a.LP = e->LP; a.last = 0; a.npos = (long)B * e->L;
    const char* ev = getenv("DMAD_LAYER_STAMPS");         // development only: per-phase cycle stamps (diagnostic build)
    const bool stamps = ev && atoi(ev) != 0;
    unsigned long long* dbg = nullptr;
    const size_t nblk = (size_t)B * (e->L / kTileT);
    if (stamps) {
        HIPCHK(hipMalloc((void**)&dbg, nblk * 8 * sizeof(unsigned long long)));
        a.dbg = dbg;
    }
    hipEvent_t e0, e1;
    HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
    launch_wn_layer_bf16_p(a, B, e->f16, st, stamps);
    HIPCHK(hipEventRecord(e0, st));
    for (int i = 0; i < iters; ++i) launch_wn_layer_bf16_p(a, B, e->f16, st, stamps);
    HIPCHK(hipEventRecord(e1, st));
    HIPCHK(hipEventSynchronize(e1));
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    *ms_per_launch = ms / iters;
    if (dbg) {      // diagnostic: mean phase lengths in shader cycles (s_memtime), printed to stderr
        std::vector<unsigned long long> h(nblk * 8);
        HIPCHK(hipMemcpy(h.data(), dbg, h.size() * 8, hipMemcpyDeviceToHost));
        double sum[9] = {0};
        for (size_t i = 0; i < 256 && i < nblk; ++i)   // per-workgroup phase sums over all its tiles
            for (int k = 0; k < 8; ++k) sum[k + 1] += (double)h[i * 8 + k];
        const size_t nwg = nblk < 256 ? nblk : 256;
        fprintf(stderr, "[dmad stamps] top-wait %.0f | gemm1 %.0f | gate0+barrier %.0f | gate||gemm2 %.0f | barrier %.0f | epilogue %.0f  (mean cycles per tile, %zu tiles); "
                "in-kernel clock %.3f GHz (shader cycles / 100 MHz ticks over the workgroups' lifetimes)\n",
                sum[1] / nblk, sum[2] / nblk, sum[3] / nblk, sum[4] / nblk, sum[5] / nblk, sum[8] / nblk, nblk,
                sum[7] > 0 ? sum[6] / sum[7] * 0.1 : 0.0);
        (void)nwg;
        (void)hipFree(dbg);
    }
    return 0;
}

}  // extern "C"

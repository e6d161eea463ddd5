// Step programs (include/mfhip.h, "step programs"): a recorded sequence of this library's own entry points, replayed without Python.
// Host code only: the file parser, the buffer table, one thunk per replayable entry, and the three model-level entries of
// SURVEY.md section 8(b) (mf_denoise_step_fused, mf_unet_forward, mf_brushnet_forward) on top of them.
// The writer is reflecting_reality_amd/program.py (Recorder.save); both sides are pinned by tests/test_program_gpu.py.
#include <string.h>

#include <string>
#include <vector>

#include "mf_common.h"

namespace {

enum ArgKind : uint32_t { A_I32 = 0, A_I64 = 1, A_F32 = 2, A_PTR = 3, A_DESC = 4 };

struct Arg {
    uint32_t kind;
    int32_t buf;       // A_PTR: buffer index (-1: null); A_DESC: index into Call::descs
    int64_t i;         // integer value / byte offset inside the buffer
    float f;
};
struct Fixup { uint32_t field_off; int32_t buf; int64_t off; };
struct Desc { std::vector<uint8_t> bytes; std::vector<Fixup> fix; };
struct Call { int fn; uint32_t stream; std::string name; std::vector<Arg> args; std::vector<Desc> descs; };
enum { F_SYNC_RECORD = -2, F_SYNC_WAIT = -3 };      // "@record" / "@wait": an event of the program recorded on / awaited by one of its streams
struct Buffer { int32_t kind; int64_t bytes; int64_t data_off; std::string name; void* ptr; };

enum Fn {
    F_GEMM_CONV, F_GROUPNORM, F_LAYERNORM, F_SOFTMAX_ROWS, F_ATTN_BF16, F_ATTN_F16, F_ATTN_F16X3, F_ATTN_F16X3_LSE, F_SPLIT_HALVES, F_QUANT_FP8,
    F_PACK_NHWC, F_UNPACK_NCHW, F_ADD, F_CAST_BF16, F_GEGLU, F_TIMESTEP_EMB, F_SILU_F32, F_CFG_DDIM_DEV, F_CFG_COMBINE, F_VAE_SAMPLE,
    F_NEAREST, F_TRANSPOSE, F_TRANSPOSE_BF16, F_TRANSPOSE_BF16_BF16, F_MEMCPY2D, F_MEMSET, F_COUNT
};
// name, argument kinds before the stream (p pointer, i int32, l int64, f float, d descriptor): program.SIGNATURES holds the same table
const struct { const char* name; const char* sig; } kFns[F_COUNT] = {
    {"mf_gemm_conv", "d"}, {"mf_groupnorm", "d"}, {"mf_layernorm", "pipipplif"}, {"mf_softmax_rows", "ppilii"},
    {"mf_attention_bf16", "plplplpliiiiif"}, {"mf_attention_f16", "plplplpliiiiif"}, {"mf_attention_f16x3", "pplpplpplpliiiiif"},
    {"mf_attention_f16x3_lse", "pplpplpplplpiiiiif"},
    {"mf_split_halves", "pppl"}, {"mf_quantize_rows_fp8", "pipplippf"}, {"mf_pack_nhwc", "pipipiiii"}, {"mf_unpack_nchw", "pilpiii"},
    {"mf_add", "pipipil"}, {"mf_cast_bf16", "ppl"}, {"mf_geglu", "pipili"}, {"mf_timestep_embedding", "ppiiif"}, {"mf_silu_f32", "ppl"},
    {"mf_cfg_ddim_step_dev", "ppfpppifl"}, {"mf_cfg_combine", "ppfpl"}, {"mf_vae_sample", "pilppiiif"}, {"mf_nearest_resize", "ppiiiii"},
    {"mf_transpose", "ppiiillll"}, {"mf_transpose_bf16", "ppiiillll"}, {"mf_transpose_bf16_bf16", "ppiiillll"},
    {"mf_memcpy2d", "plplll"}, {"mf_memset", "pil"},
};

struct Reader {
    const uint8_t* p; const uint8_t* end; bool ok = true;
    template <typename T> T get() {
        T v{};
        if (p + sizeof(T) > end) { ok = false; return v; }
        memcpy(&v, p, sizeof(T));
        p += sizeof(T);
        return v;
    }
    const uint8_t* take(size_t n) {
        const size_t padded = (n + 7) / 8 * 8;
        if (p + padded > end) { ok = false; return nullptr; }
        const uint8_t* q = p;
        p += padded;
        return q;
    }
};

}  // namespace

struct mf_program {
    std::vector<Buffer> buffers;
    std::vector<Call> calls;
    std::string meta;
    uint32_t nstreams = 1, nevents = 0;
    std::vector<hipStream_t> side;       // streams 1 .. nstreams-1 of a program recorded on several (stream 0 is the caller's)
    std::vector<hipEvent_t> events;
    int device = -1;
};

extern "C" int mf_memcpy2d(void* dst, int64_t dpitch, const void* src, int64_t spitch, int64_t width_bytes, int64_t height, void* stream) {
    MF_CHECK_ARG(dst && src && width_bytes >= 0 && height >= 0, "mf_memcpy2d: bad arguments");
    if (width_bytes == 0 || height == 0) return MF_OK;
    hipError_t e;
    if (height == 1 || (dpitch == width_bytes && spitch == width_bytes))
        e = hipMemcpyAsync(dst, src, (size_t)width_bytes * (size_t)height, hipMemcpyDeviceToDevice, (hipStream_t)stream);
    else {
        MF_CHECK_ARG(dpitch >= width_bytes && spitch >= width_bytes, "mf_memcpy2d: pitch smaller than the row");
        e = hipMemcpy2DAsync(dst, (size_t)dpitch, src, (size_t)spitch, (size_t)width_bytes, (size_t)height, hipMemcpyDeviceToDevice, (hipStream_t)stream);
    }
    if (e != hipSuccess) { mf_set_error("mf_memcpy2d: %s", hipGetErrorString(e)); return MF_ELAUNCH; }
    return MF_OK;
}

extern "C" int mf_memset(void* dst, int32_t value, int64_t bytes, void* stream) {
    MF_CHECK_ARG(dst && bytes >= 0, "mf_memset: bad arguments");
    if (bytes == 0) return MF_OK;
    const hipError_t e = hipMemsetAsync(dst, value, (size_t)bytes, (hipStream_t)stream);
    if (e != hipSuccess) { mf_set_error("mf_memset: %s", hipGetErrorString(e)); return MF_ELAUNCH; }
    return MF_OK;
}

extern "C" int mf_program_load(const void* blob, int64_t bytes, mf_program** out) {
    MF_CHECK_ARG(blob && out && bytes >= 40, "mf_program_load: null or short blob");
    Reader r{(const uint8_t*)blob, (const uint8_t*)blob + bytes};
    MF_CHECK_ARG(memcmp(r.p, "MFPROG1\0", 8) == 0, "mf_program_load: not a step program (magic)");
    r.p += 8;
    const uint32_t abi = r.get<uint32_t>(), nbuf = r.get<uint32_t>(), ncall = r.get<uint32_t>(), meta_len = r.get<uint32_t>();
    const int64_t head_len = r.get<int64_t>();
    (void)r.get<int64_t>();
    const uint32_t nstreams = r.get<uint32_t>(), nevents = r.get<uint32_t>();
    MF_CHECK_ARG(abi == MF_ABI_VERSION, "mf_program_load: the program was recorded against ABI %u, this library is ABI %d (descriptor layouts may differ): export it again",
                 abi, MF_ABI_VERSION);
    MF_CHECK_ARG(head_len <= bytes, "mf_program_load: the blob holds %lld bytes of a %lld-byte header", (long long)bytes, (long long)head_len);
    MF_CHECK_ARG(nstreams >= 1 && nstreams <= 16 && nevents <= 65536, "mf_program_load: %u streams / %u events", nstreams, nevents);
    mf_program* p = new mf_program();
    p->nstreams = nstreams;
    p->nevents = nevents;
    const uint8_t* m = r.take(meta_len);
    if (m) p->meta.assign((const char*)m, meta_len);
    for (uint32_t i = 0; i < nbuf && r.ok; ++i) {
        Buffer b{};
        b.kind = (int32_t)r.get<uint32_t>();
        const uint32_t nl = r.get<uint32_t>();
        b.bytes = r.get<int64_t>();
        b.data_off = r.get<int64_t>();
        const uint8_t* n = r.take(nl);
        if (n) b.name.assign((const char*)n, nl);
        b.ptr = nullptr;
        p->buffers.push_back(b);
    }
    for (uint32_t c = 0; c < ncall && r.ok; ++c) {
        Call call{};
        const uint32_t nl = r.get<uint32_t>(), nargs = r.get<uint32_t>();
        call.stream = r.get<uint32_t>();
        (void)r.get<uint32_t>();
        const uint8_t* n = r.take(nl);
        if (!n) break;
        call.name.assign((const char*)n, nl);
        call.fn = -1;
        for (int f = 0; f < F_COUNT; ++f)
            if (call.name == kFns[f].name) call.fn = f;
        if (call.stream >= nstreams) {
            mf_set_error("mf_program_load: call %u (%s) on stream %u of %u", c, call.name.c_str(), call.stream, nstreams);
            delete p;
            return MF_EINVAL;
        }
        if (call.name == "@record" || call.name == "@wait") {
            Arg arg{};
            arg.kind = r.get<uint32_t>();
            arg.buf = r.get<int32_t>();
            arg.i = r.get<int64_t>();
            if (nargs != 1 || arg.kind != A_I32 || arg.i < 0 || arg.i >= (int64_t)nevents) {
                mf_set_error("mf_program_load: call %u: malformed %s", c, call.name.c_str());
                delete p;
                return MF_EINVAL;
            }
            call.fn = call.name == "@record" ? F_SYNC_RECORD : F_SYNC_WAIT;
            call.args.push_back(arg);
            p->calls.push_back(std::move(call));
            continue;
        }
        if (call.fn < 0 || strlen(kFns[call.fn].sig) != nargs) {
            mf_set_error("mf_program_load: call %u: entry %s with %u arguments has no replay thunk in this library", c, call.name.c_str(), nargs);
            delete p;
            return MF_EINVAL;
        }
        std::vector<std::pair<int32_t, int64_t>> pending;      // (nfix, desc_len) of the descriptor arguments, in order
        for (uint32_t a = 0; a < nargs && r.ok; ++a) {
            Arg arg{};
            arg.kind = r.get<uint32_t>();
            arg.buf = r.get<int32_t>();
            if (arg.kind == A_F32) { arg.f = r.get<float>(); (void)r.get<int32_t>(); }
            else arg.i = r.get<int64_t>();
            const char want = kFns[call.fn].sig[a];
            const bool match = (want == 'p' && arg.kind == A_PTR) || (want == 'i' && arg.kind == A_I32) || (want == 'l' && arg.kind == A_I64) ||
                               (want == 'f' && arg.kind == A_F32) || (want == 'd' && arg.kind == A_DESC);
            if (!match || (arg.kind == A_PTR && (arg.buf < -1 || arg.buf >= (int32_t)nbuf))) {
                mf_set_error("mf_program_load: call %u (%s): argument %u does not match the entry's signature", c, call.name.c_str(), a);
                delete p;
                return MF_EINVAL;
            }
            if (arg.kind == A_DESC) { pending.push_back({arg.buf, arg.i}); arg.buf = (int32_t)pending.size() - 1; }
            call.args.push_back(arg);
        }
        for (auto& pd : pending) {
            Desc d;
            const size_t want = call.fn == F_GEMM_CONV ? sizeof(mf_gemm_desc) : sizeof(mf_groupnorm_desc);
            const uint8_t* raw = r.take((size_t)pd.second);
            if (!raw || (size_t)pd.second != want) {
                mf_set_error("mf_program_load: call %u (%s): descriptor of %lld bytes, this library's is %zu", c, call.name.c_str(), (long long)pd.second, want);
                delete p;
                return MF_EINVAL;
            }
            d.bytes.assign(raw, raw + pd.second);
            for (int32_t k = 0; k < pd.first && r.ok; ++k) {
                Fixup fx{};
                fx.field_off = r.get<uint32_t>();
                fx.buf = r.get<int32_t>();
                fx.off = r.get<int64_t>();
                if (fx.field_off + 8 > d.bytes.size() || fx.buf < 0 || fx.buf >= (int32_t)nbuf) r.ok = false;
                d.fix.push_back(fx);
            }
            call.descs.push_back(std::move(d));
        }
        p->calls.push_back(std::move(call));
    }
    if (!r.ok || p->calls.size() != ncall || p->buffers.size() != nbuf) {
        mf_set_error("mf_program_load: truncated or malformed program");
        delete p;
        return MF_EINVAL;
    }
    *out = p;
    return MF_OK;
}

extern "C" void mf_program_destroy(mf_program* p) {
    if (!p) return;
    for (hipStream_t st : p->side) (void)hipStreamDestroy(st);
    for (hipEvent_t ev : p->events) (void)hipEventDestroy(ev);
    delete p;
}
extern "C" int32_t mf_program_num_buffers(const mf_program* p) { return p ? (int32_t)p->buffers.size() : 0; }
extern "C" int32_t mf_program_num_calls(const mf_program* p) { return p ? (int32_t)p->calls.size() : 0; }
extern "C" const char* mf_program_meta(const mf_program* p) { return p ? p->meta.c_str() : ""; }

extern "C" int mf_program_buffer_info(const mf_program* p, int32_t index, int32_t* kind, int64_t* bytes, int64_t* data_offset, const char** name) {
    MF_CHECK_ARG(p && index >= 0 && index < (int32_t)p->buffers.size(), "mf_program_buffer_info: index %d out of range", index);
    const Buffer& b = p->buffers[index];
    if (kind) *kind = b.kind;
    if (bytes) *bytes = b.bytes;
    if (data_offset) *data_offset = b.data_off;
    if (name) *name = b.name.c_str();
    return MF_OK;
}

extern "C" int32_t mf_program_find_buffer(const mf_program* p, const char* name) {
    if (!p || !name) return -1;
    for (size_t i = 0; i < p->buffers.size(); ++i)
        if (p->buffers[i].name == name) return (int32_t)i;
    return -1;
}

extern "C" int mf_program_bind(mf_program* p, int32_t index, void* device_ptr) {
    MF_CHECK_ARG(p && index >= 0 && index < (int32_t)p->buffers.size(), "mf_program_bind: index %d out of range", index);
    MF_CHECK_ARG(device_ptr && ((uintptr_t)device_ptr & 15) == 0, "mf_program_bind: buffer %d (%s) needs 16-byte aligned device memory", index,
                 p->buffers[index].name.c_str());
    p->buffers[index].ptr = device_ptr;
    return MF_OK;
}

#define HIP_TRY(call, what)                                                    \
    do {                                                                       \
        const hipError_t e_ = (call);                                          \
        if (e_ != hipSuccess) {                                                \
            mf_set_error("%s: %s", what, hipGetErrorString(e_));               \
            return MF_ELAUNCH;                                                 \
        }                                                                      \
    } while (0)

namespace {

inline void* resolve(const mf_program* p, int32_t buf, int64_t off) { return buf < 0 ? nullptr : (char*)p->buffers[buf].ptr + off; }

int run_call(const mf_program* prog, const Call& c, void* s) {
    const std::vector<Arg>& a = c.args;
#define P(k) resolve(prog, a[k].buf, a[k].i)
#define FP(k) ((const float*)P(k))
#define I(k) ((int32_t)a[k].i)
#define L(k) (a[k].i)
#define F(k) (a[k].f)
    switch (c.fn) {
    case F_GEMM_CONV: case F_GROUPNORM: {
        const Desc& d = c.descs[a[0].buf];
        alignas(16) uint8_t raw[sizeof(mf_gemm_desc) > sizeof(mf_groupnorm_desc) ? sizeof(mf_gemm_desc) : sizeof(mf_groupnorm_desc)];
        memcpy(raw, d.bytes.data(), d.bytes.size());
        for (const Fixup& fx : d.fix) {
            void* ptr = resolve(prog, fx.buf, fx.off);
            memcpy(raw + fx.field_off, &ptr, sizeof(void*));
        }
        if (c.fn == F_GROUPNORM) return mf_groupnorm((const mf_groupnorm_desc*)raw, s);
        // the HOST out-fields (which row block the partial sums got, whether per-group sums were written) were read when the pass was
        // recorded — the consumer's descriptor carries the answer; the launch still wants somewhere to write them
        mf_gemm_desc* g = (mf_gemm_desc*)raw;
        int32_t rows_unused = 0, grouped_unused = 0;
        if (g->gn_part) { g->gn_part_rows = &rows_unused; g->gn_grouped = &grouped_unused; }
        if (g->defer_reduce) g->deferred_splits = &grouped_unused;      // (the recorded consumer already expects the slabs)
        return mf_gemm_conv(g, s);
    }
    case F_LAYERNORM: return mf_layernorm(P(0), I(1), P(2), I(3), FP(4), FP(5), L(6), I(7), F(8), s);
    case F_SOFTMAX_ROWS: return mf_softmax_rows(FP(0), P(1), I(2), L(3), I(4), I(5), s);
    case F_ATTN_BF16: return mf_attention_bf16(P(0), L(1), P(2), L(3), P(4), L(5), P(6), L(7), I(8), I(9), I(10), I(11), I(12), F(13), s);
    case F_ATTN_F16: return mf_attention_f16(P(0), L(1), P(2), L(3), P(4), L(5), P(6), L(7), I(8), I(9), I(10), I(11), I(12), F(13), s);
    case F_ATTN_F16X3:
        return mf_attention_f16x3(P(0), P(1), L(2), P(3), P(4), L(5), P(6), P(7), L(8), (float*)P(9), L(10), I(11), I(12), I(13), I(14), I(15), F(16), s);
    case F_ATTN_F16X3_LSE:
        return mf_attention_f16x3_lse(P(0), P(1), L(2), P(3), P(4), L(5), P(6), P(7), L(8), (float*)P(9), L(10), (float*)P(11), I(12), I(13), I(14), I(15),
                                      I(16), F(17), s);
    case F_SPLIT_HALVES: return mf_split_halves(FP(0), P(1), P(2), L(3), s);
    case F_QUANT_FP8: return mf_quantize_rows_fp8(P(0), I(1), P(2), (float*)P(3), L(4), I(5), FP(6), FP(7), F(8), s);
    case F_PACK_NHWC: return mf_pack_nhwc(FP(0), I(1), FP(2), I(3), P(4), I(5), I(6), I(7), I(8), s);
    case F_UNPACK_NCHW: return mf_unpack_nchw(P(0), I(1), L(2), (float*)P(3), I(4), I(5), I(6), s);
    case F_ADD: return mf_add(P(0), I(1), P(2), I(3), P(4), I(5), L(6), s);
    case F_CAST_BF16: return mf_cast_bf16(FP(0), P(1), L(2), s);
    case F_GEGLU: return mf_geglu(P(0), I(1), P(2), I(3), L(4), I(5), s);
    case F_TIMESTEP_EMB: return mf_timestep_embedding(FP(0), (float*)P(1), I(2), I(3), I(4), F(5), s);
    case F_SILU_F32: return mf_silu_f32(FP(0), (float*)P(1), L(2), s);
    case F_CFG_DDIM_DEV: return mf_cfg_ddim_step_dev(FP(0), FP(1), F(2), FP(3), (float*)P(4), FP(5), I(6), F(7), L(8), s);
    case F_CFG_COMBINE: return mf_cfg_combine(FP(0), FP(1), F(2), (float*)P(3), L(4), s);
    case F_VAE_SAMPLE: return mf_vae_sample(P(0), I(1), L(2), FP(3), (float*)P(4), I(5), I(6), I(7), F(8), s);
    case F_NEAREST: return mf_nearest_resize(FP(0), (float*)P(1), I(2), I(3), I(4), I(5), I(6), s);
    case F_TRANSPOSE: return mf_transpose(FP(0), (float*)P(1), I(2), I(3), I(4), L(5), L(6), L(7), L(8), s);
    case F_TRANSPOSE_BF16: return mf_transpose_bf16(FP(0), P(1), I(2), I(3), I(4), L(5), L(6), L(7), L(8), s);
    case F_TRANSPOSE_BF16_BF16: return mf_transpose_bf16_bf16(P(0), P(1), I(2), I(3), I(4), L(5), L(6), L(7), L(8), s);
    case F_MEMCPY2D: return mf_memcpy2d(P(0), L(1), P(2), L(3), L(4), L(5), s);
    case F_MEMSET: return mf_memset(P(0), I(1), L(2), s);
    default: break;
    }
#undef P
#undef FP
#undef I
#undef L
#undef F
    mf_set_error("mf_program_run: entry %s has no thunk", c.name.c_str());
    return MF_EINVAL;
}

// bind an io buffer by name when the caller passed memory for it; a program without that buffer refuses the argument
int bind_io(mf_program* p, const char* entry, const char* name, const void* ptr) {
    if (!ptr) return MF_OK;
    const int32_t i = mf_program_find_buffer(p, name);
    MF_CHECK_ARG(i >= 0 && p->buffers[i].kind == MF_PROGRAM_IO, "%s: the program has no io buffer \"%s\" (was it exported for this entry?)", entry, name);
    return mf_program_bind(p, i, const_cast<void*>(ptr));
}

}  // namespace

extern "C" int mf_program_run(mf_program* p, void* stream) {
    MF_CHECK_ARG(p, "mf_program_run: null program");
    for (size_t i = 0; i < p->buffers.size(); ++i)
        MF_CHECK_ARG(p->buffers[i].ptr, "mf_program_run: buffer %zu (%s, %lld bytes) is not bound", i, p->buffers[i].name.c_str(), (long long)p->buffers[i].bytes);
    if (p->side.size() + 1 < p->nstreams || p->events.size() < p->nevents) {
        // the program's own streams and events, on the device that is current at the first run (the only objects the library owns)
        HIP_TRY(hipGetDevice(&p->device), "mf_program_run");
        while (p->side.size() + 1 < p->nstreams) {
            hipStream_t st;
            HIP_TRY(hipStreamCreateWithFlags(&st, hipStreamNonBlocking), "mf_program_run(stream)");
            p->side.push_back(st);
        }
        while (p->events.size() < p->nevents) {
            hipEvent_t ev;
            HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming), "mf_program_run(event)");
            p->events.push_back(ev);
        }
    }
    for (size_t i = 0; i < p->calls.size(); ++i) {
        const Call& c = p->calls[i];
        void* s = c.stream == 0 ? stream : (void*)p->side[c.stream - 1];
        if (c.fn == F_SYNC_RECORD) { HIP_TRY(hipEventRecord(p->events[c.args[0].i], (hipStream_t)s), "mf_program_run(record)"); continue; }
        if (c.fn == F_SYNC_WAIT) { HIP_TRY(hipStreamWaitEvent((hipStream_t)s, p->events[c.args[0].i], 0), "mf_program_run(wait)"); continue; }
        const int rc = run_call(p, c, s);
        if (rc != MF_OK) return rc;          // (the failing entry has set mf_last_error)
    }
    return MF_OK;
}

extern "C" int mf_denoise_step_fused(mf_program* step, void* latents, const void* coef4, const void* temb_unet, const void* temb_brushnet, void* stream) {
    MF_CHECK_ARG(step, "mf_denoise_step_fused: null program");
    int rc;
    if ((rc = bind_io(step, "mf_denoise_step_fused", "latents", latents)) != MF_OK) return rc;
    if ((rc = bind_io(step, "mf_denoise_step_fused", "coef4", coef4)) != MF_OK) return rc;
    if ((rc = bind_io(step, "mf_denoise_step_fused", "temb_unet", temb_unet)) != MF_OK) return rc;
    if ((rc = bind_io(step, "mf_denoise_step_fused", "temb_brushnet", temb_brushnet)) != MF_OK) return rc;
    return mf_program_run(step, stream);
}

extern "C" int mf_unet_forward(mf_program* unet, const void* sample, const void* temb, const void* const* residuals_in, int32_t n_residuals,
                               void* eps_out, void* stream) {
    MF_CHECK_ARG(unet && n_residuals >= 0 && (n_residuals == 0 || residuals_in), "mf_unet_forward: bad arguments");
    int rc;
    if ((rc = bind_io(unet, "mf_unet_forward", "sample", sample)) != MF_OK) return rc;
    if ((rc = bind_io(unet, "mf_unet_forward", "temb", temb)) != MF_OK) return rc;
    if ((rc = bind_io(unet, "mf_unet_forward", "eps", eps_out)) != MF_OK) return rc;
    for (int32_t i = 0; i < n_residuals; ++i) {
        char name[32];
        snprintf(name, sizeof(name), "residual.%d", i);
        if ((rc = bind_io(unet, "mf_unet_forward", name, residuals_in[i])) != MF_OK) return rc;
    }
    return mf_program_run(unet, stream);
}

extern "C" int mf_brushnet_forward(mf_program* brushnet, const void* sample, const void* temb, const void* cond, void* const* residuals_out,
                                   int32_t n_residuals, void* stream) {
    MF_CHECK_ARG(brushnet && n_residuals >= 0 && (n_residuals == 0 || residuals_out), "mf_brushnet_forward: bad arguments");
    int rc;
    if ((rc = bind_io(brushnet, "mf_brushnet_forward", "sample", sample)) != MF_OK) return rc;
    if ((rc = bind_io(brushnet, "mf_brushnet_forward", "temb", temb)) != MF_OK) return rc;
    if ((rc = bind_io(brushnet, "mf_brushnet_forward", "cond", cond)) != MF_OK) return rc;
    for (int32_t i = 0; i < n_residuals; ++i) {
        char name[32];
        snprintf(name, sizeof(name), "residual.%d", i);
        if ((rc = bind_io(brushnet, "mf_brushnet_forward", name, residuals_out[i])) != MF_OK) return rc;
    }
    return mf_program_run(brushnet, stream);
}

extern "C" int mf_vae_decode(mf_program* vae_decoder, const void* z, void* image_out, void* stream) {
    MF_CHECK_ARG(vae_decoder, "mf_vae_decode: null program");
    int rc;
    if ((rc = bind_io(vae_decoder, "mf_vae_decode", "z", z)) != MF_OK) return rc;
    if ((rc = bind_io(vae_decoder, "mf_vae_decode", "image", image_out)) != MF_OK) return rc;
    return mf_program_run(vae_decoder, stream);
}

extern "C" int mf_vae_encode_moments(mf_program* vae_encoder, const void* image, void* moments_out, void* stream) {
    MF_CHECK_ARG(vae_encoder, "mf_vae_encode_moments: null program");
    int rc;
    if ((rc = bind_io(vae_encoder, "mf_vae_encode_moments", "image", image)) != MF_OK) return rc;
    if ((rc = bind_io(vae_encoder, "mf_vae_encode_moments", "moments", moments_out)) != MF_OK) return rc;
    return mf_program_run(vae_encoder, stream);
}

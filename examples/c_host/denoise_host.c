/* A denoise loop with no Python in the process: loads a step program exported by
 *   pipe.export_denoise_step("step.mfprog", ...)        (reflecting_reality_amd/pipeline.py, program.py)
 * gives every buffer device memory, and calls mf_denoise_step_fused (include/mfhip.h) once per timestep — the loop body of the
 * reference's pipelines/brushnet/pipeline_brushnet.py:1250-1332 behind one C entry.  Between steps the host copies the step's rows
 * of the schedule's tables (DDIM coefficients, the two time-embedding tables: named constants of the file) into the io buffers,
 * which is all the reference's loop does on the host besides launching.
 *
 *   gcc -O2 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iinclude examples/c_host/denoise_host.c \
 *       -Lreflecting-reality_amd/lib -lmfhip -L/opt/rocm/lib -lamdhip64 -o denoise_host
 *   LD_LIBRARY_PATH=reflecting-reality_amd/lib:/opt/rocm/lib ./denoise_host step.mfprog [latents_in.bin] [latents_out.bin] [--graph]
 *                                                                           [--prompt bind.mfprog prompt_embeds.bin]
 *
 * --prompt: a second program (program.export_bind_prompt) computes the cross-attention K / V^T of NEW prompt embeddings ([2B][77][C] in the
 * model's storage dtype) into the constants the step reads — the two files name those buffers alike, the host binds them to the same memory.
 *
 * latents_in.bin: the initial noise (NCHW fp32, the io buffer's size); without it the loop starts from the latents the file holds
 * (those before the recorded step).  --graph: capture the program's launches into a hipGraph once and replay it per step (measured
 * SLOWER than the plain calls at BASELINE configs[1], 16.4 vs 15.1 ms per step: in the graph this process captures the two branches of
 * the step do not overlap, although the same capture made from torch's streams does — DESIGN.md section 5c). */
#include <hip/hip_runtime_api.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "mfhip.h"

#define HIP_OK(call)                                                                              \
    do {                                                                                          \
        hipError_t e_ = (call);                                                                   \
        if (e_ != hipSuccess) {                                                                   \
            fprintf(stderr, "%s:%d: %s -> %s\n", __FILE__, __LINE__, #call, hipGetErrorString(e_)); \
            return 1;                                                                             \
        }                                                                                         \
    } while (0)
#define MF_OKAY(call)                                                                 \
    do {                                                                              \
        if ((call) != MF_OK) {                                                        \
            fprintf(stderr, "%s:%d: %s -> %s\n", __FILE__, __LINE__, #call, mf_last_error()); \
            return 1;                                                                 \
        }                                                                             \
    } while (0)

static void* device_buffer(mf_program* p, const char* name, int64_t* bytes_out, void** all) {
    const int32_t i = mf_program_find_buffer(p, name);
    if (i < 0) return NULL;
    int64_t bytes = 0;
    mf_program_buffer_info(p, i, NULL, &bytes, NULL, NULL);
    if (bytes_out) *bytes_out = bytes;
    return all[i];
}

int main(int argc, char** argv) {
    if (argc < 2) {
        fprintf(stderr, "usage: %s step.mfprog [latents_in.bin] [latents_out.bin] [--graph]\n", argv[0]);
        return 2;
    }
    const char* in_path = NULL;
    const char* out_path = NULL;
    const char* bind_path = NULL;
    const char* prompt_path = NULL;
    int use_graph = 0;
    for (int a = 2; a < argc; ++a) {
        if (!strcmp(argv[a], "--graph")) use_graph = 1;
        else if (!strcmp(argv[a], "--prompt") && a + 2 < argc) { bind_path = argv[a + 1]; prompt_path = argv[a + 2]; a += 2; }
        else if (!in_path) in_path = argv[a];
        else out_path = argv[a];
    }
    FILE* f = fopen(argv[1], "rb");
    if (!f) { perror(argv[1]); return 1; }
    unsigned char head[40];
    if (fread(head, 1, 40, f) != 40) { fprintf(stderr, "short file\n"); return 1; }
    int64_t head_len;
    memcpy(&head_len, head + 24, 8);
    void* blob = malloc((size_t)head_len);
    fseek(f, 0, SEEK_SET);
    if (fread(blob, 1, (size_t)head_len, f) != (size_t)head_len) { fprintf(stderr, "short header\n"); return 1; }
    mf_program* prog = NULL;
    MF_OKAY(mf_program_load(blob, head_len, &prog));
    free(blob);
    HIP_OK(hipSetDevice(0));
    const int32_t nbuf = mf_program_num_buffers(prog);
    void** dev = (void**)calloc((size_t)nbuf, sizeof(void*));
    int64_t total[3] = {0, 0, 0};
    for (int32_t i = 0; i < nbuf; ++i) {
        int32_t kind; int64_t bytes, off; const char* name;
        MF_OKAY(mf_program_buffer_info(prog, i, &kind, &bytes, &off, &name));
        HIP_OK(hipMalloc(&dev[i], (size_t)(bytes > 0 ? bytes : 16)));
        if (off >= 0 && bytes > 0) {
            void* host = malloc((size_t)bytes);
            fseek(f, (long)off, SEEK_SET);
            if (fread(host, 1, (size_t)bytes, f) != (size_t)bytes) { fprintf(stderr, "short data for buffer %s\n", name); return 1; }
            HIP_OK(hipMemcpy(dev[i], host, (size_t)bytes, hipMemcpyHostToDevice));
            free(host);
        }
        MF_OKAY(mf_program_bind(prog, i, dev[i]));
        total[kind] += bytes;
    }
    fclose(f);
    printf("program: %d calls, %d buffers (%.1f MB constants, %.1f MB workspace, %.3f MB io)\nmeta: %s\n", mf_program_num_calls(prog), nbuf,
           total[MF_PROGRAM_CONST] / 1e6, total[MF_PROGRAM_WORKSPACE] / 1e6, total[MF_PROGRAM_IO] / 1e6, mf_program_meta(prog));
    int64_t lat_bytes = 0, coef_bytes = 0, tu_bytes = 0, tb_bytes = 0, tab_c = 0, tab_u = 0, tab_b = 0;
    void* lat = device_buffer(prog, "latents", &lat_bytes, dev);
    void* coef = device_buffer(prog, "coef4", &coef_bytes, dev);
    void* tu = device_buffer(prog, "temb_unet", &tu_bytes, dev);
    void* tb = device_buffer(prog, "temb_brushnet", &tb_bytes, dev);
    char* table_c = (char*)device_buffer(prog, "table.coef4", &tab_c, dev);
    char* table_u = (char*)device_buffer(prog, "table.temb_unet", &tab_u, dev);
    char* table_b = (char*)device_buffer(prog, "table.temb_brushnet", &tab_b, dev);
    if (!lat || !coef || !tu || !tb || !table_c || !table_u || !table_b) {
        fprintf(stderr, "not a denoise-step program (io buffers latents / coef4 / temb_* and their tables)\n");
        return 1;
    }
    const int steps = (int)(tab_c / coef_bytes);
    if (in_path) {
        FILE* g = fopen(in_path, "rb");
        void* host = malloc((size_t)lat_bytes);
        if (!g || fread(host, 1, (size_t)lat_bytes, g) != (size_t)lat_bytes) { fprintf(stderr, "%s: need %lld bytes of latents\n", in_path, (long long)lat_bytes); return 1; }
        fclose(g);
        HIP_OK(hipMemcpy(lat, host, (size_t)lat_bytes, hipMemcpyHostToDevice));
        free(host);
    }
    hipStream_t stream;
    HIP_OK(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
    if (bind_path) {
        /* the prompt-binding program: buffers the step program also names (weights, the K / V^T it writes) share the step's memory */
        FILE* bf = fopen(bind_path, "rb");
        if (!bf || fread(head, 1, 40, bf) != 40) { fprintf(stderr, "%s: cannot read\n", bind_path); return 1; }
        memcpy(&head_len, head + 24, 8);
        blob = malloc((size_t)head_len);
        fseek(bf, 0, SEEK_SET);
        if (fread(blob, 1, (size_t)head_len, bf) != (size_t)head_len) { fprintf(stderr, "%s: short header\n", bind_path); return 1; }
        mf_program* bind = NULL;
        MF_OKAY(mf_program_load(blob, head_len, &bind));
        free(blob);
        int shared = 0;
        for (int32_t i = 0; i < mf_program_num_buffers(bind); ++i) {
            int32_t kind; int64_t bytes, off; const char* name;
            MF_OKAY(mf_program_buffer_info(bind, i, &kind, &bytes, &off, &name));
            const int32_t j = kind == MF_PROGRAM_WORKSPACE ? -1 : mf_program_find_buffer(prog, name);
            void* mem = NULL;
            if (j >= 0) { mem = dev[j]; ++shared; }
            else {
                HIP_OK(hipMalloc(&mem, (size_t)(bytes > 0 ? bytes : 16)));
                if (off >= 0 && bytes > 0) {
                    void* host = malloc((size_t)bytes);
                    fseek(bf, (long)off, SEEK_SET);
                    if (fread(host, 1, (size_t)bytes, bf) != (size_t)bytes) { fprintf(stderr, "short data for buffer %s\n", name); return 1; }
                    HIP_OK(hipMemcpy(mem, host, (size_t)bytes, hipMemcpyHostToDevice));
                    free(host);
                }
            }
            if (!strcmp(name, "prompt_embeds")) {
                FILE* pf = fopen(prompt_path, "rb");
                void* host = malloc((size_t)bytes);
                if (!pf || fread(host, 1, (size_t)bytes, pf) != (size_t)bytes) { fprintf(stderr, "%s: need %lld bytes of prompt embeddings\n", prompt_path, (long long)bytes); return 1; }
                fclose(pf);
                HIP_OK(hipMemcpy(mem, host, (size_t)bytes, hipMemcpyHostToDevice));
                free(host);
            }
            MF_OKAY(mf_program_bind(bind, i, mem));
        }
        fclose(bf);
        MF_OKAY(mf_program_run(bind, stream));
        HIP_OK(hipStreamSynchronize(stream));
        printf("prompt bound: %d calls, %d buffers shared with the step program\n", mf_program_num_calls(bind), shared);
        mf_program_destroy(bind);
    }
    hipGraphExec_t exec = NULL;
    if (use_graph) {
        /* a first eager run (nothing lazy is left to initialise, but it keeps the capture free of first-use work), on a copy of the latents */
        void* keep;
        HIP_OK(hipMalloc(&keep, (size_t)lat_bytes));
        HIP_OK(hipMemcpyAsync(keep, lat, (size_t)lat_bytes, hipMemcpyDeviceToDevice, stream));
        MF_OKAY(mf_program_run(prog, stream));
        HIP_OK(hipMemcpyAsync(lat, keep, (size_t)lat_bytes, hipMemcpyDeviceToDevice, stream));
        HIP_OK(hipStreamSynchronize(stream));
        hipGraph_t graph;
        HIP_OK(hipStreamBeginCapture(stream, hipStreamCaptureModeThreadLocal));
        MF_OKAY(mf_program_run(prog, stream));
        HIP_OK(hipStreamEndCapture(stream, &graph));
        HIP_OK(hipGraphInstantiateWithFlags(&exec, graph, hipGraphInstantiateFlagAutoFreeOnLaunch));
        HIP_OK(hipFree(keep));
    }
    hipEvent_t e0, e1;
    HIP_OK(hipEventCreate(&e0));
    HIP_OK(hipEventCreate(&e1));
    HIP_OK(hipEventRecord(e0, stream));
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (int i = in_path ? 0 : 1; i < steps; ++i) {
        HIP_OK(hipMemcpyAsync(coef, table_c + (size_t)i * coef_bytes, (size_t)coef_bytes, hipMemcpyDeviceToDevice, stream));
        HIP_OK(hipMemcpyAsync(tu, table_u + (size_t)i * tu_bytes, (size_t)tu_bytes, hipMemcpyDeviceToDevice, stream));
        HIP_OK(hipMemcpyAsync(tb, table_b + (size_t)i * tb_bytes, (size_t)tb_bytes, hipMemcpyDeviceToDevice, stream));
        if (exec) HIP_OK(hipGraphLaunch(exec, stream));
        else MF_OKAY(mf_denoise_step_fused(prog, NULL, NULL, NULL, NULL, stream));      /* NULL: the bindings made above stay */
    }
    HIP_OK(hipEventRecord(e1, stream));
    clock_gettime(CLOCK_MONOTONIC, &t1);          /* everything is enqueued; the device may still be many steps behind */
    HIP_OK(hipStreamSynchronize(stream));
    float ms = 0.0f;
    HIP_OK(hipEventElapsedTime(&ms, e0, e1));
    const int ran = steps - (in_path ? 0 : 1);
    float* host = (float*)malloc((size_t)lat_bytes);
    HIP_OK(hipMemcpy(host, lat, (size_t)lat_bytes, hipMemcpyDeviceToHost));
    double sum = 0.0, sabs = 0.0;
    for (int64_t i = 0; i < lat_bytes / 4; ++i) { sum += host[i]; sabs += host[i] < 0 ? -host[i] : host[i]; }
    const double enq_ms = (t1.tv_sec - t0.tv_sec) * 1e3 + (t1.tv_nsec - t0.tv_nsec) * 1e-6;
    printf("%d denoise steps (%s): %.3f ms = %.3f ms per step on the device, %.3f ms per step of host time to enqueue; latents sum %.6f mean|x| %.6f\n", ran,
           exec ? "hipGraph replay of the program" : "mf_denoise_step_fused", ms, ran ? ms / ran : 0.0f, ran ? enq_ms / ran : 0.0, sum, sabs / (double)(lat_bytes / 4));
    if (out_path) {
        FILE* g = fopen(out_path, "wb");
        if (!g || fwrite(host, 1, (size_t)lat_bytes, g) != (size_t)lat_bytes) { perror(out_path); return 1; }
        fclose(g);
    }
    free(host);
    mf_program_destroy(prog);
    for (int32_t i = 0; i < nbuf; ++i) hipFree(dev[i]);
    free(dev);
    return 0;
}

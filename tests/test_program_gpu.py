"""Step programs (reflecting_reality_amd/program.py, csrc/program.hip, include/mfhip.h "step programs"): the C-ABI call sequence of
a whole pass, recorded once and replayed by the library without the models — mf_denoise_step_fused, mf_unet_forward,
mf_brushnet_forward of SURVEY.md section 8(b).  The bar is bit-exactness against the Python-sequenced pass on the same inputs
(the replay launches the same kernels with the same descriptors), for the loop body of pipeline_brushnet.py:1250-1332 and for
the two networks on their own."""
import ctypes as C
import json
import struct

import pytest
import torch

pytestmark = pytest.mark.gpu

from reflecting_reality_amd import hip, program, synth  # noqa: E402
from test_pipeline_gpu import _tiny_pipe  # noqa: E402

DEV = "cuda"


def _call_args(inp, steps, noise, **kw):
    args = dict(prompt_embeds=inp["prompt_embeds"], negative_prompt_embeds=inp["negative_prompt_embeds"], image=inp["image"],
                mask=inp["mask"], depth=inp["depth"], num_inference_steps=steps, guidance_scale=7.5, latents=inp["latents"].clone(),
                output_type="latent", height=16, width=32, conditioning_noise=noise)
    args.update(kw)
    return args


@pytest.mark.parametrize("prec", ["fp32", "bf16", "fp16", "f16x3"])
def test_denoise_step_program_replays_the_loop_bit_exactly(prec, tmp_path):
    """Export step 1 of a 5-step run, then drive ALL five steps from the file alone (tables -> io buffers -> mf_program_run):
    the latents after every step equal the pipeline's own (hipGraph replay), bit for bit."""
    pipe = _tiny_pipe(prec)
    inp = synth.pipeline_inputs(2, 16, 32, seed=7, cross_dim=32, vae_scale=2)
    noise = torch.randn(4, 4, 8, 16, generator=torch.Generator().manual_seed(3))
    per_step = []

    def grab(p, i, t, kw):
        per_step.append(kw["latents"].detach().float().cpu().clone())
        return {}
    ref = pipe(**_call_args(inp, 5, noise, callback_on_step_end=grab, callback_on_step_end_tensor_inputs=["latents"])).images.float().cpu()
    assert len(per_step) == 5 and torch.equal(per_step[-1], ref)
    pipe._graph_state = None
    path = str(tmp_path / "step.mfprog")
    info = pipe.export_denoise_step(path, **_call_args(inp, 5, noise))
    assert torch.equal(info["result"].images.float().cpu(), ref), "the exporting run itself must not change the result"
    assert info["calls"] > 20 and "mf_gemm_conv" in info["entries"] and "mf_cfg_ddim_step_dev" in info["entries"]
    assert info["streams"] >= 2 and "@record" in info["entries"] and "@wait" in info["entries"], "the capture's BrushNet || UNet fork is part of the program"
    print(f"[{prec}] program: {info['calls']} calls on {info['streams']} streams ({info['events']} events), {info['buffers']} buffers, {info['bytes'] / 1e6:.1f} MB file "
          f"({info['const_bytes'] / 1e6:.1f} MB constants, {info['workspace_bytes'] / 1e6:.1f} MB workspace), entries {info['entries']}")
    del pipe
    prog = program.Program(path, DEV)
    meta = json.loads(prog.meta)
    assert meta["entry"] == "mf_denoise_step_fused" and meta["steps"] == 5 and prog.num_calls == info["calls"]
    # (1) the file as it is reproduces the recorded step: latents after step 0 -> latents after step 1
    assert torch.equal(prog.buffer("latents", torch.float32).view(per_step[0].shape).cpu(), per_step[0])
    prog.run()
    torch.cuda.synchronize()
    assert torch.equal(prog.buffer("latents", torch.float32).view(per_step[1].shape).cpu(), per_step[1])
    # (2) the whole loop through mf_denoise_step_fused with the HOST's own io buffers
    lib = hip.load()
    lat = inp["latents"].to(DEV).float().contiguous()
    coef = prog.buffer("table.coef4", torch.float32).view(5, 4)
    tu = prog.buffer("table.temb_unet", torch.float32).view(5, -1)
    tb = prog.buffer("table.temb_brushnet", torch.float32).view(5, -1)
    c_cur, tu_cur, tb_cur = torch.empty_like(coef[0]), torch.empty_like(tu[0]), torch.empty_like(tb[0])
    for i in range(5):
        c_cur.copy_(coef[i]); tu_cur.copy_(tu[i]); tb_cur.copy_(tb[i])
        hip._check(lib.mf_denoise_step_fused(prog._h, C.c_void_p(lat.data_ptr()), C.c_void_p(c_cur.data_ptr()), C.c_void_p(tu_cur.data_ptr()),
                                             C.c_void_p(tb_cur.data_ptr()), hip._stream()), "mf_denoise_step_fused")
        torch.cuda.synchronize()
        assert torch.equal(lat.cpu(), per_step[i]), f"step {i}: the replayed program differs from the pipeline by {(lat.cpu() - per_step[i]).abs().max()}"
    # (3) the replay is capturable: one hipGraph of the program, replayed
    lat.copy_(inp["latents"].to(DEV).float())
    c_cur.copy_(coef[0]); tu_cur.copy_(tu[0]); tb_cur.copy_(tb[0])
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        prog.run()
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(lat.cpu(), per_step[0])
    prog.close()


@pytest.mark.parametrize("prec", ["fp32", "bf16"])
def test_unet_and_brushnet_programs_match_the_models(prec, tmp_path):
    """mf_brushnet_forward writes the 28 residuals the model returns; mf_unet_forward, fed with them, the model's noise prediction."""
    pipe = _tiny_pipe(prec)
    unet, bn = pipe.unet, pipe.brushnet
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 4, 8, 16, generator=g).to(DEV)
    cond = torch.randn(2, 5, 8, 16, generator=g).to(DEV)
    ehs = torch.randn(2, 7, 32, generator=g).to(DEV)
    tvals = torch.tensor([801.0, 401.0], device=DEV)
    tab_b, tab_u = bn.time_embedding_table(tvals, 2), unet.time_embedding_table(tvals, 2)
    temb_b, temb_u = tab_b[0].clone(), tab_u[0].clone()
    t = torch.zeros(1, device=DEV)
    down, mid, up = bn(x, t, encoder_hidden_states=None, brushnet_cond=cond, conditioning_scale=1.0, return_dict=False, _temb=temb_b)
    res = list(down) + [mid] + list(up)
    eps = unet(x, t, encoder_hidden_states=ehs, down_block_add_samples=list(down), mid_block_add_sample=mid, up_block_add_samples=list(up),
               return_dict=False, _temb=temb_u)[0]
    ref_res = [r.clone() for r in res]
    pb, pu = str(tmp_path / "brushnet.mfprog"), str(tmp_path / "unet.mfprog")
    ib = program.export_brushnet(bn, pb, x, temb_b, cond)
    iu = program.export_unet(unet, pu, x, temb_u, ehs, down, mid, up)
    assert ib["meta"]["residuals"] == len(res) == iu["meta"]["residuals"]
    lib = hip.load()
    # new inputs: another timestep's rows, other latents — the programs are functions of their io buffers
    x2 = torch.randn(2, 4, 8, 16, generator=g).to(DEV)
    cond2 = torch.randn(2, 5, 8, 16, generator=g).to(DEV)
    temb_b2, temb_u2 = tab_b[1].clone(), tab_u[1].clone()
    down2, mid2, up2 = bn(x2, t, encoder_hidden_states=None, brushnet_cond=cond2, conditioning_scale=1.0, return_dict=False, _temb=temb_b2)
    res2 = [r.clone() for r in list(down2) + [mid2] + list(up2)]
    eps2 = unet(x2, t, encoder_hidden_states=ehs, down_block_add_samples=list(down2), mid_block_add_sample=mid2, up_block_add_samples=list(up2),
                return_dict=False, _temb=temb_u2)[0].clone()
    progb, progu = program.Program(pb, DEV), program.Program(pu, DEV)
    outs = [torch.empty_strided(r.shape, r.stride(), dtype=r.dtype, device=DEV) for r in ref_res]
    ptrs = (C.c_void_p * len(outs))(*[o.data_ptr() for o in outs])
    for xx, tt, cc, want in ((x, temb_b, cond, ref_res), (x2, temb_b2, cond2, res2)):
        hip._check(lib.mf_brushnet_forward(progb._h, C.c_void_p(xx.data_ptr()), C.c_void_p(tt.data_ptr()), C.c_void_p(cc.data_ptr()), ptrs, len(outs),
                                           hip._stream()), "mf_brushnet_forward")
        torch.cuda.synchronize()
        for i, (o, w) in enumerate(zip(outs, want)):
            assert torch.equal(o, w), f"residual {i} differs by {(o.float() - w.float()).abs().max()}"
    eps_out = torch.empty_like(eps)
    for xx, tt, want_res, want in ((x, temb_u, ref_res, eps), (x2, temb_u2, res2, eps2)):
        rp = (C.c_void_p * len(want_res))(*[r.data_ptr() for r in want_res])
        hip._check(lib.mf_unet_forward(progu._h, C.c_void_p(xx.data_ptr()), C.c_void_p(tt.data_ptr()), rp, len(want_res), C.c_void_p(eps_out.data_ptr()),
                                       hip._stream()), "mf_unet_forward")
        torch.cuda.synchronize()
        assert torch.equal(eps_out, want), f"eps differs by {(eps_out - want).abs().max()}"
    # an entry called on a program that was not exported for it is refused, with the buffer it misses
    rc = lib.mf_denoise_step_fused(progu._h, C.c_void_p(x.data_ptr()), None, None, None, hip._stream())
    assert rc != 0 and b"latents" in lib.mf_last_error()
    progb.close(); progu.close()


def test_recorder_refuses_what_it_cannot_replay():
    """No partial exports: a torch kernel that is not a copy / fill, an entry without a thunk, a launch on another stream."""
    a = torch.ones(64, device=DEV)
    b = torch.ones(64, device=DEV)
    hip.load()
    with pytest.raises(program.ProgramError, match="aten::add"):
        with program.Recorder(dict(a=a)):
            _ = a + b
    assert hip._RECORDER is None
    with pytest.raises(program.ProgramError, match="no replay thunk"):
        with program.Recorder(dict(a=a)):
            hip.silu_bwd(a, b)
    assert hip._RECORDER is None
    side = torch.cuda.Stream()
    with pytest.raises(program.ProgramError, match="another stream"):
        with program.Recorder(dict(a=a)):
            with torch.cuda.stream(side):
                hip.silu_f32(a)
    # copies, fills and concatenations ARE recorded (as mf_memcpy2d / mf_memset) and replay
    with program.Recorder(dict(a=a)) as rec:
        c = torch.cat([a, b])
        d = c.view(2, 64)[:, :16].contiguous()            # a strided copy: 2 rows of 64 bytes, 256 apart
        e = torch.zeros(8, device=DEV)
        hip.silu_f32(d.view(-1))
        del c, e
    names = [c[0] for c in rec.resolved]
    assert names.count("mf_memcpy2d") == 3 and names.count("mf_memset") == 1 and names[-1] == "mf_silu_f32"


def test_program_file_is_validated(tmp_path):
    a = torch.arange(64, device=DEV, dtype=torch.float32)
    with program.Recorder(dict(x=a)) as rec:
        y = hip.silu_f32(a)
        rec.output("y", y)
        want = y.clone()
    path = str(tmp_path / "p.mfprog")
    rec.save(path, meta="{}")
    lib = hip.load()
    blob = bytearray(open(path, "rb").read())
    h = C.c_void_p()
    assert lib.mf_program_load(bytes(blob), C.c_int64(len(blob)), C.byref(h)) == 0
    assert lib.mf_program_run(h, hip._stream()) != 0 and b"not bound" in lib.mf_last_error()
    lib.mf_program_destroy(h)
    bad = bytearray(blob)
    bad[8:12] = struct.pack("<I", hip.ABI_VERSION - 1)
    assert lib.mf_program_load(bytes(bad), C.c_int64(len(bad)), C.byref(h)) != 0 and b"ABI" in lib.mf_last_error()
    bad = bytearray(blob)
    bad[0] = ord("X")
    assert lib.mf_program_load(bytes(bad), C.c_int64(len(bad)), C.byref(h)) != 0 and b"magic" in lib.mf_last_error()
    head_len = struct.unpack("<q", blob[24:32])[0]
    assert lib.mf_program_load(bytes(blob[:head_len - 8]), C.c_int64(head_len - 8), C.byref(h)) != 0
    prog = program.Program(path, DEV)
    prog.run()
    torch.cuda.synchronize()
    assert torch.equal(prog.buffer("y", torch.float32), want)
    prog.buffer("x", torch.float32).mul_(2.0)
    prog.run()
    torch.cuda.synchronize()
    assert torch.equal(prog.buffer("y", torch.float32), hip.silu_f32(a * 2.0))
    prog.close()


@pytest.mark.parametrize("graph", [False, True, "prompt"])
def test_c_host_runs_the_loop_without_python(graph, tmp_path):
    """examples/c_host/denoise_host.c: a C program (gcc, the HIP runtime, libmfhip — no Python, no torch in the process) loads the
    exported step, runs all five steps from the initial noise and writes the latents the pipeline produces, bit for bit."""
    import os
    import shutil
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    gcc = shutil.which("gcc")
    if gcc is None or not os.path.exists("/opt/rocm/include/hip/hip_runtime_api.h"):
        pytest.skip("no gcc / ROCm headers on this machine")
    exe = str(tmp_path / "denoise_host")
    libdir = os.path.join(root, "reflecting-reality_amd", "lib")
    subprocess.run([gcc, "-O2", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", f"-I{os.path.join(root, 'include')}",
                    os.path.join(root, "examples", "c_host", "denoise_host.c"), f"-L{libdir}", "-lmfhip", "-L/opt/rocm/lib", "-lamdhip64", "-o", exe],
                   check=True, capture_output=True, text=True)
    pipe = _tiny_pipe("bf16")
    inp = synth.pipeline_inputs(2, 16, 32, seed=7, cross_dim=32, vae_scale=2)
    noise = torch.randn(4, 4, 8, 16, generator=torch.Generator().manual_seed(3))
    path = str(tmp_path / "step.mfprog")
    info = pipe.export_denoise_step(path, **_call_args(inp, 5, noise))
    ref = info["result"].images.float().cpu()
    lat_in, lat_out = str(tmp_path / "in.bin"), str(tmp_path / "out.bin")
    inp["latents"].float().contiguous().numpy().tofile(lat_in)
    extra = ["--graph"] if graph is True else []
    if graph == "prompt":        # another prompt through the prompt-binding program: the step file stays the one exported with the first
        other = synth.pipeline_inputs(2, 16, 32, seed=8, cross_dim=32, vae_scale=2)
        bind_path, pe_path = str(tmp_path / "bind.mfprog"), str(tmp_path / "prompt.bin")
        program.export_bind_prompt(pipe.unet, bind_path)
        kw = _call_args(inp, 5, noise)
        kw.update(prompt_embeds=other["prompt_embeds"], negative_prompt_embeds=other["negative_prompt_embeds"])
        ref = pipe(**kw).images.float().cpu()
        pe = torch.cat([other["negative_prompt_embeds"], other["prompt_embeds"]]).to(torch.bfloat16).contiguous()
        pe.view(torch.int16).numpy().tofile(pe_path)
        extra = ["--prompt", bind_path, pe_path]
    env = dict(os.environ, LD_LIBRARY_PATH=f"{libdir}:/opt/rocm/lib:" + os.environ.get("LD_LIBRARY_PATH", ""))
    out = subprocess.run([exe, path, lat_in, lat_out] + extra, capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    print(out.stdout)
    import numpy as np
    got = torch.from_numpy(np.fromfile(lat_out, dtype=np.float32)).view(ref.shape)
    assert torch.equal(got, ref), f"the C host's latents differ from the pipeline's by {(got - ref).abs().max()}"


@pytest.mark.parametrize("prec", ["fp32", "bf16"])
def test_vae_programs_match_the_model(prec, tmp_path):
    """mf_vae_decode / mf_vae_encode_moments replay AutoencoderKL.decode / the encoder up to the posterior's moments, bit for bit, on
    inputs other than the recorded ones."""
    vae = _tiny_pipe(prec).vae
    g = torch.Generator().manual_seed(9)
    z, z2 = (torch.randn(2, 4, 8, 16, generator=g).to(DEV) for _ in range(2))
    x, x2 = (torch.randn(2, 3, 16, 32, generator=g).to(DEV) for _ in range(2))
    pd, pe = str(tmp_path / "dec.mfprog"), str(tmp_path / "enc.mfprog")
    idec, ienc = program.export_vae_decode(vae, pd, z), program.export_vae_encode(vae, pe, x)
    print(f"[{prec}] decode: {idec['calls']} calls {idec['entries']}; encode: {ienc['calls']} calls")
    lib = hip.load()
    dec, enc = program.Program(pd, DEV), program.Program(pe, DEV)
    for zz, xx in ((z, x), (z2, x2)):
        want_img = vae.decode(zz, return_dict=False)[0]
        img = torch.empty_like(want_img)
        hip._check(lib.mf_vae_decode(dec._h, C.c_void_p(zz.data_ptr()), C.c_void_p(img.data_ptr()), hip._stream()), "mf_vae_decode")
        want_m = vae._moments(xx)
        m = torch.empty_like(want_m)
        hip._check(lib.mf_vae_encode_moments(enc._h, C.c_void_p(xx.data_ptr()), C.c_void_p(m.data_ptr()), hip._stream()), "mf_vae_encode_moments")
        torch.cuda.synchronize()
        assert torch.equal(img, want_img) and torch.equal(m, want_m)
    dec.close(); enc.close()


@pytest.mark.parametrize("prec", ["bf16", "fp8"])
def test_sdxl_denoise_step_program(prec, tmp_path):
    """The SDXL pipeline's step (pipeline_brushnet_sd_xl.py:1497-1600: added text_time conditioning, BrushNet-XL, linear
    projections; fp8 Linears with per-row quantisation in the fp8 mode) exports and replays like SD1.5's."""
    from oracle import mirrorfusion_ref as R
    from reflecting_reality_amd import DDIMScheduler, StableDiffusionXLBrushNetPipeline
    from test_xl_gpu import SD_SCHED, build_xl
    try:
        unet, bn, vae = build_xl(prec)
    except Exception as e:           # (the tiny XL widths may not meet the fp8 kernels' multiples)
        pytest.skip(f"tiny SDXL does not build in {prec}: {e}")
    pipe = StableDiffusionXLBrushNetPipeline(vae=vae, text_encoder=None, text_encoder_2=None, tokenizer=None, tokenizer_2=None, unet=unet,
                                             brushnet=bn, scheduler=DDIMScheduler(**SD_SCHED, clip_sample=False))
    pipe.set_progress_bar_config(disable=True)
    inp = synth.pipeline_inputs(1, 16, 16, seed=99, cross_dim=48, vae_scale=2)
    gp = torch.Generator().manual_seed(100)
    pooled, npooled = torch.randn(1, 24, generator=gp), torch.randn(1, 24, generator=gp)
    noise = torch.randn(2, 4, 8, 8, generator=gp)
    kw = dict(prompt_embeds=inp["prompt_embeds"], negative_prompt_embeds=inp["negative_prompt_embeds"], pooled_prompt_embeds=pooled,
              negative_pooled_prompt_embeds=npooled, image=inp["image"], mask=inp["mask"], num_inference_steps=4, guidance_scale=5.0,
              output_type="latent", brushnet_conditioning_scale=1.0, height=16, width=16, original_size=(24, 20), crops_coords_top_left=(2, 1),
              target_size=(16, 16), conditioning_noise=noise)
    ref = pipe(latents=inp["latents"].clone(), **kw).images.float().cpu()
    pipe._graph_state = None
    path = str(tmp_path / "xl_step.mfprog")
    info = pipe.export_denoise_step(path, latents=inp["latents"].clone(), **kw)
    assert torch.equal(info["result"].images.float().cpu(), ref)
    print(f"[sdxl {prec}] {info['calls']} calls on {info['streams']} streams, entries {info['entries']}")
    prog = program.Program(path, DEV)
    lat = prog.buffer("latents", torch.float32)
    lat.copy_(inp["latents"].to(DEV).float().reshape(-1))
    coef, tu, tb = (prog.buffer(n, torch.float32).view(4, -1) for n in ("table.coef4", "table.temb_unet", "table.temb_brushnet"))
    for i in range(4):
        prog.buffer("coef4", torch.float32).copy_(coef[i]); prog.buffer("temb_unet", torch.float32).copy_(tu[i]); prog.buffer("temb_brushnet", torch.float32).copy_(tb[i])
        prog.run()
    torch.cuda.synchronize()
    assert torch.equal(lat.view(ref.shape).cpu(), ref)
    prog.close()


def test_bind_prompt_program_changes_the_prompt_of_an_exported_step(tmp_path):
    """export_bind_prompt: the K / V^T a denoise-step program reads as constants are written by a second program from the host's prompt
    embeddings — the step exported with prompt A, run with prompt B, equals the pipeline's run with prompt B."""
    pipe = _tiny_pipe("bf16")
    inp = synth.pipeline_inputs(2, 16, 32, seed=7, cross_dim=32, vae_scale=2)
    other = synth.pipeline_inputs(2, 16, 32, seed=8, cross_dim=32, vae_scale=2)
    noise = torch.randn(4, 4, 8, 16, generator=torch.Generator().manual_seed(3))
    pstep, pbind = str(tmp_path / "step.mfprog"), str(tmp_path / "bind.mfprog")
    pipe.export_denoise_step(pstep, **_call_args(inp, 5, noise))
    ib = program.export_bind_prompt(pipe.unet, pbind)
    assert ib["calls"] == 2 * ib["meta"]["layers"]
    kw = _call_args(inp, 5, noise)
    kw.update(prompt_embeds=other["prompt_embeds"], negative_prompt_embeds=other["negative_prompt_embeds"])
    ref_b = pipe(**kw).images.float().cpu()
    ref_a = pipe(**_call_args(inp, 5, noise)).images.float().cpu()
    assert not torch.equal(ref_a, ref_b)
    del pipe
    step = program.Program(pstep, DEV)
    bind = program.Program(pbind, DEV, share=step)

    def loop():
        step.buffer("latents", torch.float32).copy_(inp["latents"].to(DEV).float().reshape(-1))
        coef, tu, tb = (step.buffer(n, torch.float32).view(5, -1) for n in ("table.coef4", "table.temb_unet", "table.temb_brushnet"))
        for i in range(5):
            step.buffer("coef4", torch.float32).copy_(coef[i]); step.buffer("temb_unet", torch.float32).copy_(tu[i]); step.buffer("temb_brushnet", torch.float32).copy_(tb[i])
            step.run()
        torch.cuda.synchronize()
        return step.buffer("latents", torch.float32).view(ref_a.shape).cpu().clone()
    assert torch.equal(loop(), ref_a)
    pe_b = torch.cat([other["negative_prompt_embeds"], other["prompt_embeds"]]).to(DEV, torch.bfloat16).contiguous()
    bind.buffer("prompt_embeds", torch.bfloat16).copy_(pe_b.reshape(-1))
    bind.run()
    assert torch.equal(loop(), ref_b), "the step exported with prompt A, re-bound to prompt B, differs from the pipeline's run with B"
    bind.close(); step.close()


@pytest.mark.parametrize("sched", ["pndm", "unipc"])
def test_step_program_with_a_multistep_scheduler_returns_the_noise_prediction(sched, tmp_path):
    """PNDM / UniPC (the schedulers the reference's scripts select: pipeline default, test_brushnet.py:158) keep host-side state, so their
    exported step ends with the guided noise prediction ("eps"); the host's scheduler steps from it.  Driving the program with the
    Python scheduler reproduces the pipeline's latents bit for bit."""
    from reflecting_reality_amd import PNDMScheduler, UniPCMultistepScheduler
    from test_pipeline_gpu import SD_SCHED
    pipe = _tiny_pipe("bf16")
    mk = (lambda: PNDMScheduler(**SD_SCHED, skip_prk_steps=True)) if sched == "pndm" else (lambda: UniPCMultistepScheduler(**{k: v for k, v in SD_SCHED.items() if k != "set_alpha_to_one"}))
    pipe.scheduler = mk()
    inp = synth.pipeline_inputs(2, 16, 32, seed=7, cross_dim=32, vae_scale=2)
    noise = torch.randn(4, 4, 8, 16, generator=torch.Generator().manual_seed(3))
    path = str(tmp_path / "step.mfprog")
    info = pipe.export_denoise_step(path, **_call_args(inp, 6, noise))
    ref = info["result"].images.float().cpu()
    assert "eps" in info["meta"]["result"] and "mf_cfg_combine" in info["entries"] and "mf_cfg_ddim_step_dev" not in info["entries"]
    prog = program.Program(path, DEV)
    s = mk()
    s.set_timesteps(6, device=DEV)
    lat = inp["latents"].to(DEV).float() * s.init_noise_sigma
    tu, tb = (prog.buffer(n, torch.float32) for n in ("table.temb_unet", "table.temb_brushnet"))
    ts = s.timesteps
    tu, tb = tu.view(len(ts), -1), tb.view(len(ts), -1)
    for i, t in enumerate(ts):
        prog.buffer("latents", torch.float32).copy_(lat.reshape(-1))
        prog.buffer("temb_unet", torch.float32).copy_(tu[i]); prog.buffer("temb_brushnet", torch.float32).copy_(tb[i])
        prog.run()
        eps = prog.buffer("eps", torch.float32).view(lat.shape).clone()
        lat = s.step(eps, t, lat, return_dict=False)[0]
    torch.cuda.synchronize()
    assert torch.equal(lat.float().cpu(), ref), f"{sched}: program + host scheduler differ from the pipeline by {(lat.float().cpu() - ref).abs().max()}"
    prog.close()

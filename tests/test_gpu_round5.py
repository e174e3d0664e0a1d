"""GPU (-m gpu), round 5: the parity-suite gaps VERDICT r4 named -- the lock-step batch AT the BASELINE config-4 shape (n = 4 clips of
180 x 320, where the launcher leaves the one-round kernel selection: 4-wave conv kernels, un-fused conv pairs) in both storage modes,
and the SURVEY 8(b) threading contract ("safe to call from multiple host threads on different streams") exercised from two real host
threads, followed by crfp_shutdown() and a fresh call."""
import os
import threading

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

T = torch.from_numpy
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def dev():
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def orc():
    from oracle import crfp_oracle
    return crfp_oracle


@pytest.fixture(autouse=True)
def _nograd():
    with torch.no_grad():
        yield


def _engine(storage="f32", seed=7):
    from crfp_amd import synth
    from crfp_amd.engine import DSVEngine
    sd = synth.make_state_dict(seed)
    return DSVEngine({k: T(v.copy()) for k, v in sd.items()}, dev(), storage=storage), sd


def _stats(got, ref):
    d = (got.detach().cpu().double() - ref.double()).abs()
    mse = float((d ** 2).mean())
    return float(d.max()), float(d.mean()), (99.0 if mse == 0 else -10 * np.log10(mse))


# ------------------------------------------------------------------------------------------------ config 4, lock-step, full shape
@pytest.mark.parametrize("storage", ["bf16", "f32"])
def test_config4_lockstep_batch_at_the_real_shape(orc, storage):
    """BASELINE configs[3] as one rank runs it since round 4: 4 independent 180 x 320 clips in ONE crfp_dsv_forward_batch call.  At this
    size N * tiles exceeds one round of the chip (512 workgroup slots), so the launcher picks the multi-round kernels -- bf16: the
    4-wave conv kernel instead of conv3x3_bf16x8 and two launches instead of the fused conv pairs; fp32: 4 x 450 tiles of the 8-wave
    kernel -- a selection the small-map bit-identity tests (tests/test_gpu_round4.py, N * tiles <= 512) never reach.  t = 3 keeps
    the steady-state frame (warp + DCN + carried features) in the test twice.
    (a) lock-step == the loop of one-clip calls, bit for bit, on both schedules; (b) clip 0 against the oracle (fp32: the reference's
    arithmetic, 2e-4; bf16: the builder-defined storage twin with the yardstick of tests/test_gpu_bf16.py)."""
    from crfp_amd import benchutil, synth
    eng, sd = _engine(storage)
    seeds = benchutil.rank_clip_seeds(0, 4)
    clips = [synth.make_clip(s, 1, 3, 180, 320, fv_size=96, sigma_t=10.0) for s in seeds]
    lrs, fvs, mks = (T(np.concatenate([c[k] for c in clips], axis=0)).to(dev()) for k in range(3))
    eng.batch_mode = "loop"
    ref = eng.forward(lrs, fvs, mks).clone()
    torch.cuda.synchronize()
    assert torch.isfinite(ref).all() and not eng.overflowed()
    for single in (False, True):
        eng.single_stream = single
        eng.batch_mode = "lockstep"
        got = eng.forward(lrs, fvs, mks)
        torch.cuda.synchronize()
        assert torch.equal(got, ref), f"{storage}: lock-step batch != one-clip calls at 4 x 3 x 180 x 320 (single_stream={single}): " \
                                      f"max diff {float((got - ref).abs().max()):.3e}"
    assert not eng.overflowed()
    P = orc.load_numpy_state(sd)
    l0, f0, m0 = (T(a) for a in clips[0])
    if storage == "f32":
        want = orc.crfp_dsv_forward(P, l0, f0, m0)
        mx, mean, psnr = _stats(got[:1], want)
        print(f"config-4 shape fp32, clip 0 of the lock-step batch vs oracle: max {mx:.2e} mean {mean:.2e}")
        assert mx < 2e-4, mx
    else:
        with orc.bf16_storage():
            want = orc.crfp_dsv_forward(orc.bf16_weights(P), l0, f0, m0)
        mx, mean, psnr = _stats(got[:1], want)
        print(f"config-4 shape bf16, clip 0 of the lock-step batch vs twin: max {mx:.2e} mean {mean:.2e} PSNR {psnr:.1f} dB")
        assert mean <= 3e-4 and mx <= 2e-2 and psnr >= 65.0, (mx, mean, psnr)


# ------------------------------------------------------------------------------------------------ two host threads
def test_two_host_threads_on_their_own_streams_match_sequential_calls():
    """SURVEY 8(b) "Threading/streams": the library keeps per-thread side-stream tables (engine.hip side_table()) and promises that two
    host threads may call it at the same time on different streams.  Two threading.Thread workers, each with its OWN torch stream and
    its OWN DSVEngine (fp32 and bf16 -- both twins of the engine at once), interleave clip forwards (two-stream schedule: each call
    forks onto that thread's side stream) with streamed frames for several rounds; every result must equal the same call made
    sequentially from the main thread, bit for bit.  Then crfp_shutdown() (all threads' streams destroyed) and a fresh call."""
    from crfp_amd import _lib, synth
    d = dev()
    L = _lib.lib()
    cfg = [("f32", 91, (2, 3, 36, 64)), ("bf16", 92, (1, 4, 27, 45))]
    engs, clips, want_clip, want_stream = [], [], [], []
    for storage, seed, (n, t, h, w) in cfg:
        eng, _ = _engine(storage)
        cs = [synth.make_clip(seed + 10 * i, 1, t, h, w, fv_size=64) for i in range(n)]
        c = tuple(T(np.concatenate([x[k] for x in cs], axis=0)).to(d) for k in range(3))
        engs.append(eng)
        clips.append(c)
        want_clip.append(eng.forward(*c).clone())
        eng.clear_states()
        want_stream.append(torch.stack([eng.stream_frame(c[0][0, i], c[1][0, i], c[2][0, i]).clone() for i in range(c[0].shape[1])]))
        eng.clear_states()
    torch.cuda.synchronize()
    tables0 = L.crfp_debug_side_tables()
    errors, barrier = [], threading.Barrier(2)

    def worker(k):
        try:
            torch.cuda.set_device(d)
            st = torch.cuda.Stream(device=d)
            eng, (lrs, fvs, mks) = engs[k], clips[k]
            with torch.no_grad(), torch.cuda.stream(st):
                for rnd in range(4):
                    barrier.wait(timeout=120)          # both threads enter the library together, every round
                    a = eng.forward(lrs, fvs, mks).clone()
                    eng.clear_states()
                    frames = []
                    for i in range(lrs.shape[1]):
                        frames.append(eng.stream_frame(lrs[0, i], fvs[0, i], mks[0, i]).clone())
                        if i == 1:
                            b = eng.forward(lrs, fvs, mks).clone()   # a clip call in the middle of the streamed sequence (own workspace)
                    st.synchronize()
                    if not torch.equal(a, want_clip[k]) or not torch.equal(b, want_clip[k]):
                        errors.append(f"thread {k} round {rnd}: clip forward differs from the sequential call")
                    if not torch.equal(torch.stack(frames), want_stream[k]):
                        errors.append(f"thread {k} round {rnd}: streamed frames differ from the sequential calls")
                    if eng.overflowed():
                        errors.append(f"thread {k} round {rnd}: status word set")
        except Exception as e:   # noqa: BLE001 -- reported by the main thread
            errors.append(f"thread {k}: {type(e).__name__}: {e}")
            try:
                barrier.abort()
            except Exception:
                pass

    threads = [threading.Thread(target=worker, args=(k,)) for k in range(2)]
    for th in threads:
        th.start()
    for th in threads:
        th.join(timeout=600)
    assert not any(th.is_alive() for th in threads), "worker threads hung"
    assert not errors, errors
    torch.cuda.synchronize()
    # each worker leased one table; after they exited the tables went back to the free list, so a second generation of threads reuses them
    tables1 = L.crfp_debug_side_tables()
    assert tables0 <= tables1 <= tables0 + 2, (tables0, tables1)
    gen2 = [threading.Thread(target=worker, args=(k,)) for k in range(2)]
    barrier.reset()
    for th in gen2:
        th.start()
    for th in gen2:
        th.join(timeout=600)
    assert not errors, errors
    torch.cuda.synchronize()
    assert L.crfp_debug_side_tables() == tables1, "exited threads' side-stream tables were not reused"
    # shutdown with nothing in flight, then the library must come back by itself
    assert L.crfp_shutdown() == 0
    for k in range(2):
        assert torch.equal(engs[k].forward(*clips[k]), want_clip[k]), "first call after crfp_shutdown() differs"
    torch.cuda.synchronize()

#!/usr/bin/env python3
"""Drives the HOST side of libhx_mi355's entry points on a box WITHOUT a GPU, against the AddressSanitizer host-only build
(make -C hirl4ucav_amd/csrc asan-host; tools/asan_host.sh sets HX_LIBRARY and preloads the sanitizer runtime).  The C ABI takes device pointers it
never dereferences on the host, so made-up addresses stand in for device memory; what runs under the sanitizer is everything between the entry point
and its launches: argument checks, launch-shape choices at every size class, job packing into the kernel-argument structs, the multi-launch sequences
of hx_hirl_learn* / hx_hirl_front + hx_hirl_learn_back / hx_sac_learn, the host-side pointer arrays of the peer-read exchanges, the error strings.
In that build a failed launch is not an error (HX_HOST_DRYRUN), so 0 means "every host stage ran"; argument errors must still come back as errors.
Refuses to run where a GPU is visible (made-up device addresses would be dereferenced by real kernels).  SURVEY.md 5 (sanitizer target)."""
import ctypes
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def main():
    import torch

    if torch.cuda.is_available():
        print("asan_host_drive: a GPU is visible — this driver passes made-up device addresses and is for the CPU box only")
        return 2
    from hirl4ucav_amd import _lib
    from hirl4ucav_amd.agents import engine as E
    from hirl4ucav_amd.agents import sac_engine as S

    L = _lib.load()
    base = [0x7E0000000000]

    def dev(nbytes):  # a made-up, 256-byte aligned "device allocation"
        p = base[0]
        base[0] += (int(nbytes) + 255) & ~255
        return p

    ran = []

    def ok(name, *args):
        _lib.call(name, *args)
        ran.append(name)

    def refused(name, *args):
        try:
            _lib.call(name, *args)
        except _lib.HxError as e:
            ran.append(name + " (refused)")
            assert name in str(e), e
            return
        raise AssertionError(f"{name} accepted bad arguments")

    for i, cls in ((0, _lib.HxStepOpts), (1, E.HxNets), (2, E.HxHyper), (3, E.HxBatch), (4, E.HxSample), (5, S.HxSacNets), (6, S.HxSacBatch)):
        _lib.check_struct(i, cls)
    L.hx_hirl_workspace_floats.restype = L.hx_sac_workspace_floats.restype = L.hx_bf16_images_elems.restype = L.hx_actor_message_floats.restype = ctypes.c_int64
    A, C = L.hx_actor_param_count(), L.hx_critic_param_count()
    hyper = E.HxHyper(0.99, 0.005, 1e-3, 1e-3, 0.0, 0.5, 10000.0, 1, 0)

    # ---- env: reset / step at every launch-shape class, with and without the fused insert ----
    for n in (1, 200, 4096, 40000, 65536, 1 << 20):
        pitch = n + 1056 if n >= 262144 else n
        state, obs, act = dev(37 * pitch * 4), dev(n * 13 * 4), dev(n * 16)
        rew, done, succ, epi = dev(n * 4), dev(n), dev(n), dev(n * 4)
        ok("hx_env_reset", state, n, pitch, None, None, 1, 1, 7, 0, epi, obs, None)
        cap = max(2 * n, 1 << 12)
        for ring in (None, dev(cap * 128)):
            o = _lib.HxStepOpts(1500, 1, 1, 0, 7, epi, ring, dev(cap) if ring else None, cap if ring else 0, dev(8) if ring else None, dev(4096), None, None, 0)
            ok("hx_env_step", state, n, pitch, act, obs, rew, done, succ, ctypes.byref(o), None)
        ok("hx_env_rearm", state, n, pitch, None, None)
    refused("hx_env_step", None, 0, 0, None, None, None, None, None, None, None)
    ok("hx_label_transitions", dev(13 * 4000), dev(16 * 1000), dev(13 * 4000), 1000, dev(4000), dev(1000), dev(1000), None)

    # ---- acting: every format at the per-tile and the persistent sizes, alone and with the env step ----
    actor, w2f, w2x, w2b = dev(A * 4), dev(512 * 256 * 4), dev(3 * 512 * 256 * 2), dev(512 * 256 * 2)
    ok("hx_pack_w2_f32i", actor, 13, w2f, None)
    ok("hx_pack_w2_x9", actor, 13, w2x, None)
    ok("hx_pack_w2_bf16", actor, 13, w2b, None)
    for n in (16, 4096, 5120, 8192, 16384, 65536):
        state, obs, act = dev(37 * n * 4), dev(n * 13 * 4), dev(n * 16)
        rew, done, succ, epi = dev(n * 4), dev(n), dev(n), dev(n * 4)
        cap = max(2 * n, 1 << 12)
        o = _lib.HxStepOpts(1500, 1, 1, 0, 7, epi, dev(cap * 128), dev(cap), cap, dev(8), dev(4096), None, None, 0)
        for mode in (0, 3, 3 + 16):
            ok("hx_actor_act_f32i", actor, w2f, obs, n, act, mode, None, 0.1, 1, 0, 1, 0.0, None)
            ok("hx_actor_act_x9", actor, w2x, obs, n, act, mode, None, 0.1, 1, 0, 1, 0.0, None)
            ok("hx_actor_act_bf16", actor, w2b, obs, n, act, mode, None, 0.1, 1, 0, 1, 0.01, None)
        ok("hx_actor_act", actor, obs, n, act, 1, dev(16), 0.0, 1, 0, 1, 0.0, None, None)
        ok("hx_actor_act_step_f32i", actor, w2f, state, n, n, obs, act, 3, None, 0.1, 1, 0, 1, 0.0, rew, done, succ, ctypes.byref(o), None)
        ok("hx_actor_act_step_x9", actor, w2x, state, n, n, obs, act, 3, None, 0.1, 1, 0, 1, 0.0, rew, done, succ, ctypes.byref(o), None)
        ok("hx_actor_act_step_bf16", actor, w2b, state, n, n, obs, act, 3, None, 0.1, 1, 0, 1, 0.0, rew, done, succ, ctypes.byref(o), None)
    refused("hx_actor_act_x9", actor, None, dev(64), 16, dev(64), 0, None, 0.0, 1, 0, 1, 0.0, None)

    # ---- the HIRL update: one-call, staged, sampled, front + back; fp32 and bf16 images; B = 128 .. 1024 ----
    for B, bf16 in ((128, False), (256, False), (512, False), (1024, False), (128, True)):
        ws = dev(int(L.hx_hirl_workspace_floats(B)) * 4)
        images = dev(int(L.hx_bf16_images_elems()) * 2) if bf16 else None
        nets = E.HxNets(dev(A * 4), dev(C * 4), dev(A * 4), dev(C * 4), dev(A * 4), dev(A * 4), dev(C * 4), dev(A * 4), dev(A * 4), dev(C * 4), dev(C * 4),
                        dev(32), dev(4), dev(4), ws, images if bf16 else None, w2f, images, None, None if bf16 else w2x)
        batch = E.HxBatch(dev(B * 128), dev(B * 128), B, dev(16))
        nb, hb, bb = ctypes.byref(nets), ctypes.byref(hyper), ctypes.byref(batch)
        cap = 1 << 16
        smp = E.HxSample(dev(8), cap, dev(cap * 128), dev(20000 * 128), 19999, dev(20000 * 128), 20000, B - 16, 3, 1, 0.2, dev(B * 4), dev(B * 4), 0)
        if bf16:
            ok("hx_pack_update_images", nb, None)
        for actor_phase, polyak, w_kind in ((0, 0, 2), (1, 0, 1), (1, 1, 0)):
            ok("hx_hirl_learn", nb, bb, hb, 5, actor_phase, 3, polyak, w_kind, 0.3, 0.05, None)
            ok("hx_hirl_learn_sampled", nb, bb, hb, ctypes.byref(smp), 5, actor_phase, 3, polyak, w_kind, 0.3, 0.05, None)
            ok("hx_hirl_learn_sampled", nb, bb, hb, None, 5, actor_phase, 3, polyak, w_kind, 0.3, 0.05, None)
        for actor_fwd in (0, 1, 2):
            ok("hx_hirl_critic_grads", nb, bb, hb, actor_fwd, None)
            ok("hx_hirl_critic_grads_sampled", nb, bb, hb, ctypes.byref(smp), actor_fwd, None)
        ok("hx_adam", nb, hb, 0 | 16, 5, 0.5, 0, 0.0, 0.0, B, None)
        ok("hx_hirl_actor_backward", nb, bb, hb, 1, 1, None)
        ok("hx_hirl_actor_wgrad", nb, hb, B, B, 1, 0.0, 0.05, None)
        ok("hx_adam", nb, hb, 1, 3, 1.0, 1, 0.0, 0.05, B, None)
        msg = dev(int(L.hx_actor_message_floats()) * 4)
        ok("hx_hirl_actor_wgrad_split", nb, hb, B, msg, None)
        ok("hx_adam_mixed", nb, hb, 1, 3, 0.125, 1, 0.0, 0.05, 8 * B, msg, None)
        ok("hx_polyak", nb, hb, None)
        ok("hx_bc_train_actor", nb, bb, hb, 4, None)
        ok("hx_sample_batch", smp.total, cap, smp.ring, smp.expert_ring, 19999, smp.bc_table, 20000, B, B - 16, 1, 3, 1, 0.2, smp.idx, smp.idx_bc, batch.noise,
           batch.rows, batch.bc_rows, None)
        ok("hx_sample_batch_guarded", smp.total, cap, smp.ring, smp.expert_ring, 19999, smp.bc_table, 20000, B, B - 16, 1, 3, 1, 0.2, smp.idx, smp.idx_bc,
           batch.noise, batch.rows, batch.bc_rows, 4096, None)
        if B > 256:
            continue
        # the front launch at every acting role (per-tile, streaming / persistent), with and without launch C, then both back halves
        for n in ((4096, 8192, 16384, 65536) if not bf16 else (4096, 16384, 131072)):
            state, obs, act = dev(37 * n * 4), dev(n * 13 * 4), dev(n * 16)
            rew, done, succ, epi = dev(n * 4), dev(n), dev(n), dev(n * 4)
            rcap = max(2 * n, 1 << 20)
            o = _lib.HxStepOpts(1500, 1, 1, 0, 7, epi, dev(rcap * 128), dev(rcap), rcap, dev(8), dev(4096), None, None, 0)
            nxt = E.HxSample(o.total, rcap, o.ring, smp.expert_ring, 19999, smp.bc_table, 20000, B, 3, 2, 0.2, dev(B * 4), dev(B * 4), n)
            nxt_tiles = E.HxBatch(dev(B * 128), dev(B * 128), B, dev(16))
            for k, (actor_phase, with_c) in enumerate(((0, 0), (1, 0), (0, 1))):
                fr = E.HxFront(dev(256), dev(4), k + 1, with_c)
                ok("hx_hirl_front", state, n, n, obs, act, 3 | (0 if bf16 else 32), None, 0.1, 1, 0, k + 1, rew, done, succ, ctypes.byref(o), nb, bb, hb, actor_phase, 2,
                   ctypes.byref(fr), None)
                ok("hx_hirl_learn_back", nb, bb, hb, 5, actor_phase, 3, 0, 2, 0.0, 0.0, ctypes.byref(nxt), ctypes.byref(nxt_tiles), with_c, None)
                ok("hx_hirl_critic_grads_back", nb, bb, hb, ctypes.byref(nxt), ctypes.byref(nxt_tiles), with_c, None)
        refused("hx_hirl_front", state, n, n, obs, act, 3 | 32, None, 0.1, 1, 0, 1, rew, done, succ, ctypes.byref(o), nb, bb, hb, 0, 2,
                ctypes.byref(E.HxFront(dev(256), dev(4), 0, 0)), None)  # a 0-based epoch

    # ---- SAC ----
    B = 128
    P = L.hx_sac_policy_param_count()
    sn = S.HxSacNets(dev(P * 4), dev(C * 4), dev(C * 4), dev(P * 4), dev(C * 4), dev(P * 4), dev(P * 4), dev(C * 4), dev(C * 4), dev(32), dev(16),
                     dev(int(L.hx_sac_workspace_floats(B)) * 4), w2f, w2x)
    sb = S.HxSacBatch(dev(B * 128), B, None, None, 5, 1)
    cap = 1 << 16
    ssmp = E.HxSample(dev(8), cap, dev(cap * 128), None, 0, None, 0, B, 3, 1, 0.0, dev(B * 4), None, 0)
    hs = ctypes.byref(E.HxHyper(0.99, 0.005, 1e-3, 1e-3, 0.0, 0.5, 0.0, 0, 1))
    for polyak_first in (0, 1):
        ok("hx_sac_learn", ctypes.byref(sn), ctypes.byref(sb), hs, ctypes.byref(ssmp), polyak_first, 3, -4.0, None)
        ok("hx_sac_learn", ctypes.byref(sn), ctypes.byref(sb), hs, None, polyak_first, 3, -4.0, None)
        ok("hx_sac_critic_grads", ctypes.byref(sn), ctypes.byref(sb), hs, polyak_first, None)
    ok("hx_sac_policy_grads", ctypes.byref(sn), ctypes.byref(sb), hs, None)
    for which in (0, 1):
        ok("hx_sac_adam", ctypes.byref(sn), hs, which, 3, 1.0, -4.0, None)
    for n in (4096, 16384):
        state, obs, act = dev(37 * n * 4), dev(n * 13 * 4), dev(n * 16)
        ok("hx_sac_act", sn.policy, obs, n, act, 2, None, 1, 0, 1, None, None)
        ok("hx_sac_act_f32i", sn.policy, w2f, obs, n, act, 2, None, 1, 0, 1, None)
        ok("hx_sac_act_x9", sn.policy, w2x, w2f, obs, n, act, 2, None, 1, 0, 1, None)

    # ---- the exchanges' host-side pointer arrays (world 2, 3, 8) ----
    vp = ctypes.c_void_p
    for world in (2, 3, 8):
        n = 276488
        bufs = (vp * world)(*[dev(n * 4) for _ in range(world)])
        reds = (vp * world)(*[dev(n * 4) for _ in range(world)])
        flags = (vp * world)(*[dev(64) for _ in range(world)])
        flags2 = (vp * world)(*[dev(64) for _ in range(world)])
        ok("hx_allreduce_oneshot", dev(n * 4), bufs, flags, dev(4), world, world - 1, n, 1, 200, None)
        for bf16 in (0, 1):
            ok("hx_allreduce_twostage", dev(n * 4), bufs, reds, flags, flags2, dev(4), world, 0, n, 1, 200, bf16, None)
    refused("hx_allreduce_oneshot", dev(64), None, None, dev(4), 2, 0, 16, 1, 200, None)
    refused("hx_rccl_allreduce", None, None, 0, 0, None)
    refused("hx_rccl_allreduce_bf16", None, None, None, 6, None)
    sizes = (ctypes.c_int32 * 8)()
    assert L.hx_abi_sizes(sizes) == 0
    print(f"asan_host_drive: {len(ran)} calls over {len(set(ran))} entry points ran their host paths; last error string: {L.hx_last_error().decode()[:80]!r}")
    return 0


if __name__ == "__main__":
    sys.exit(main())

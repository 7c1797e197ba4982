"""Two-stream issue order for the vector step  act -> env.step (+insert) -> sample -> learn.

The reference runs these strictly one after the other (train_all.py:343-361).  Two of the dependencies are only apparent:
the TD3-style agents touch the acting network on every SECOND learn() (HIRL.py:291,332), and learn() reads the minibatch
tiles the sampler gathered, never the replay ring.  So on a critic-only learn() the next step's act + env.step can run
beside it on a second HIP stream, as soon as the sampler has read the ring — same reads, same writes, same values: the
results are bit-identical to the serial order (tests/test_hirl_gpu.py).

Measured on one MI355X at 4,096 envs, B = 128 (profiles/README.md): 145 us/step against 143 serial — act's 256 workgroups and
learn's 1024-thread workgroups want the same CUs and LDS, so the stages mostly take turns anyway and the two event hand-offs
cost what little is hidden.  bench.py therefore keeps the serial order by default (`--overlap` turns this on); the class stays
because the exchange latency of the sharded update (N > 1) is the case it can still pay for, once that is measurable.
"""
import torch


class VectorStepPipeline:
    def __init__(self, device, overlap=True):
        self.overlap = bool(overlap)
        self.main = torch.cuda.current_stream(device)
        self.side = torch.cuda.Stream(device) if self.overlap else None
        self.sampled = torch.cuda.Event()
        self.stepped = torch.cuda.Event()
        self.issued = False  # act + env.step of the coming step are already in flight

    def act_and_step(self, fn):
        """fn() enqueues act + env.step.  Skipped when the previous learn() already issued it."""
        if self.issued:
            self.issued = False
            return
        fn()

    def prefetch(self, fn, acting_net_untouched):
        """Call between sample() and learn(): if the coming learn() leaves the acting network alone, issue the NEXT
        step's act + env.step on the side stream behind the sampler."""
        if not (self.overlap and acting_net_untouched):
            return
        self.sampled.record(self.main)
        self.side.wait_event(self.sampled)
        with torch.cuda.stream(self.side):
            fn()
        self.stepped.record(self.side)
        self.issued = True

    def join(self):
        """Call after learn(): the next sampler (and anything else on the main stream) waits for the side stream."""
        if self.issued:
            self.main.wait_event(self.stepped)

"""Two-stream issue order for the vector step  act -> env.step (+insert) -> sample -> learn.

The reference runs these strictly one after the other (train_all.py:343-361).  Two of the dependencies are only apparent:
the TD3-style agents touch the acting network on every SECOND learn() (HIRL.py:291,332), and learn() reads the minibatch
tiles the sampler gathered, never the replay ring.  So on a critic-only learn() the next step's act + env.step can run
beside it on a second HIP stream, as soon as the sampler has read the ring — same reads, same writes, same values: the
results are bit-identical to the serial order (tests/test_hirl_gpu.py).

Measured on one MI355X at 4,096 envs, B = 128 (profiles/README.md): 145 us/step against 143 serial — act's 256 workgroups and
learn's 1024-thread workgroups want the same CUs and LDS, so the stages mostly take turns anyway and the two event hand-offs
cost what little is hidden (with act + env step as one launch: 107.8 vs 100.1 us).  In the sharded path the side stream can be
released at the gradient all-reduce (HirlEngine.learn(before_exchange=pipe.fire)), so that it works while the main stream WAITS for
the exchange instead of beside its compute kernels; on one GPU, where that wait is empty, this form shows what the two event
hand-offs themselves cost: 117.5 vs 100.1 us per step, i.e. ~17 us per overlapped step.  bench.py therefore keeps ONE stream by default
at any N (`--overlap` turns this on); whether hiding a 30-us launch behind an all-reduce beats that cost is a question for an 8-GPU box.
"""
import torch


class VectorStepPipeline:
    def __init__(self, device, overlap=True):
        self.overlap = bool(overlap)
        self.main = torch.cuda.current_stream(device)
        self.side = torch.cuda.Stream(device) if self.overlap else None
        self.ready = torch.cuda.Event()
        self.stepped = torch.cuda.Event()
        self.issued = False   # act + env.step of the coming step are already in flight
        self._armed = None    # the issue function, waiting for fire()

    def act_and_step(self, fn):
        """fn() enqueues act + env.step.  Skipped when the previous learn() already issued it."""
        if self.issued:
            self.issued = False
            return
        fn()

    def arm(self, fn, acting_net_untouched, engine=None):
        """Call between sample() and learn(): if the coming learn() leaves the acting network alone, the NEXT step's act + env.step
        may go out on the side stream — when fire() says so.  engine: the agent engine whose learn() follows; a draw it DEFERRED into
        that learn() (sample(defer=True)) reads the ring and its row count inside learn()'s first launch, which an env step running
        beside it would change under its workgroups — refused here."""
        if engine is not None and self.overlap and acting_net_untouched and getattr(engine, "_pending", None) is not None:
            raise RuntimeError("VectorStepPipeline.arm: the engine holds a deferred minibatch draw (sample(defer=True)); draw with defer=False "
                               "when the next env step may overlap learn()")
        self._armed = fn if (self.overlap and acting_net_untouched) else None

    def fire(self):
        """The moment to issue the armed work: everything enqueued on the main stream so far (the sampler's ring reads included) comes
        first, the rest of learn() runs beside it.  The sharded engine calls this right before its gradient all-reduce, so the side
        stream works while the main stream waits for the exchange — not while it computes."""
        fn, self._armed = self._armed, None
        if fn is None:
            return
        self.ready.record(self.main)
        self.side.wait_event(self.ready)
        with torch.cuda.stream(self.side):
            fn()
        self.stepped.record(self.side)
        self.issued = True

    def prefetch(self, fn, acting_net_untouched, engine=None):
        """arm + fire at once: the side stream starts right behind the sampler.  engine: as in arm() — pass the engine whose learn() follows,
        so that a draw it deferred into that learn() is refused here too (once fired, nothing is armed any more for learn() to notice)."""
        self.arm(fn, acting_net_untouched, engine=engine)
        self.fire()

    def join(self):
        """Call after learn(): work that nobody fired goes out now; the next sampler (and anything else on the main stream) waits
        for the side stream."""
        self.fire()
        if self.issued:
            self.main.wait_event(self.stepped)

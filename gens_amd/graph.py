"""A training step captured ONCE into a HIP graph and replayed (torch.cuda.CUDAGraph = hipGraph on ROCm).

A fused training step is ~140 launches of 5 - 1 600 us each: the GPU needs 7.4 ms for them and the host about as long to
enqueue them, so any hiccup of the host thread (a busy node, a cold page cache: the first process on a fresh box measured 12 - 30 ms per
step for the same kernels) lands in the step time.  Nothing in such a step depends on the host any more -- the point selections are
device-side index lists, the reference's mid-step errors are device flags (ImplicitSurface.check_deferred), the generator draws reach the
device through a page-locked buffer -- so the whole step (forward, loss, backward, optimiser) is a fixed sequence of launches: capture it
once, replay it with ONE launch per step.

    step = GraphedStep(body, surfaces=[model.implicit_surface], optimizer=opt)     # body(): forward + loss + backward + optimizer.step() -> loss
    for _ in range(n):
        copy the step's inputs into the tensors body() reads (rays, pixels, ... : tensor.copy_(new), same shapes)
        loss = step()                     # refreshes the host draws, replays, returns the captured loss tensor
        float(loss)                       # the read-back synchronises; step.check() then raises what the reference would have raised

Warm-up: torch.cuda.graph needs a few eager runs of body() first.  They would apply real optimiser steps and consume host generator draws, so
a graphed run would start from other parameters and another draw sequence than an eager one: the parameters, the optimiser state, the CPU
generator state and the buffers of the modules named in `modules` (BatchNorm statistics) are saved before the warm-up and put back after
it -- replay k is step k.  The whole GenS.forward training step -- MIOpen's 2-D convolutions, the 3-D U-Net, the render, the loss, backward,
Adam: ~1 000 launches -- replays in 29.6 ms against 32 - 54 ms eager, depending on the host (scripts/train_step_bench.py --full --graph).

What must hold (checked where it can be): the optimiser is capturable (torch.optim.Adam(..., capturable=True)), shapes and the set of
tensors body() touches do not change between replays, inputs are updated IN PLACE, and body() does not synchronise with the host.  The eager
path stays the reference-compatible default (runner.py calls the model step by step); this is the opt-in for loops that own their step.
"""
import torch


class GraphedStep:
    def __init__(self, body, surfaces, optimizer, warmup=3, modules=()):
        """body: callable -> scalar loss tensor, running forward + loss + backward + optimizer.step() on the current stream.  surfaces: the
        ImplicitSurface modules whose host draws the step uses.  optimizer: zero_grad(set_to_none=True) is called around the capture.
        modules: modules whose BUFFERS a step moves (BatchNorm running statistics and batch counters of a trunk in training mode): saved
        before the warm-up and put back after it like the parameters."""
        self.surfaces = list(surfaces)
        for group in optimizer.param_groups:
            assert group.get("capturable", False) or group.get("fused", False), "build the optimiser with capturable=True (its step counter must live on the device)"
        import copy
        params = [p for group in optimizer.param_groups for p in group["params"]]
        saved_params = [p.detach().clone() for p in params]
        buffers = [b for m in modules for b in m.buffers()]
        saved_buffers = [b.detach().clone() for b in buffers]
        saved_opt = copy.deepcopy(optimizer.state_dict())
        saved_rng = torch.get_rng_state()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        # The body's model calls run EAGERLY in here: with the default warmup=3 the third call would otherwise start an AutoGraph capture of its own
        # inside this warm-up, whose end_capture() takes the surface's page-locked draw / check buffers away -- the whole-step capture below would then
        # allocate page-locked memory inside an open capture -- and an AutoGraph entry with its pools would stay alive for nothing.
        with no_auto_graph(), torch.cuda.stream(side):         # (warm-up off the default stream, as torch.cuda.graph requires)
            for _ in range(warmup):
                optimizer.zero_grad(set_to_none=True)
                body()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        for s in self.surfaces:
            s.check_deferred()
        with torch.no_grad():                                  # the warm-up never happened: parameters, Adam moments / step counters, generator
            for p, q in zip(params, saved_params):
                p.copy_(q)
            for b, q in zip(buffers, saved_buffers):
                b.copy_(q)
        state_before = optimizer.state_dict()["state"]
        if saved_opt["state"]:
            optimizer.load_state_dict(saved_opt)
        else:                                                  # a fresh optimiser: its state tensors exist now (capture needs them) -- zero them in place
            for st in state_before.values():
                for v in st.values():
                    if torch.is_tensor(v):
                        v.zero_()
        torch.set_rng_state(saved_rng)
        optimizer.zero_grad(set_to_none=True)                  # the captured backward then ASSIGNS the gradients (no accumulation across replays)
        self.graph = torch.cuda.CUDAGraph()
        try:
            with no_auto_graph(), _capturing(self.graph):
                self.loss = body()
        finally:
            torch.set_rng_state(saved_rng)                     # (the capture pass drew one step's numbers without running a step)
        self._replayed = None

    def __call__(self):
        if self._replayed is not None:
            # the previous replay's copy node must have READ the page-locked draw buffer before it is overwritten (free when the caller already
            # read the loss back; a loop that does not would otherwise replay with torn or duplicated draws)
            self._replayed.synchronize()
        for s in self.surfaces:
            s.refresh_host_draws()                             # the reference's generator, its order: the graph's copy node carries them over
        self.graph.replay()
        self._replayed = torch.cuda.Event()
        self._replayed.record()
        return self.loss

    def check(self):
        """After the caller synchronised with the replay (the loss read-back): the errors the reference raises mid-step."""
        for s in self.surfaces:
            s.check_deferred_host()


# ----------------------------------------------------------------------------------------------------------------------
# The same idea behind the reference's UNCHANGED training loop (runner.py:139-197, 295-343):
#
#     outputs = model(mode, inputs, cos_anneal_ratio=..., step=...)          # replays graph A into static output buffers
#     loss = loss_fn(outputs, inputs, step)["loss"]                          # the caller's own code, eager
#     optimizer.zero_grad(); loss.backward(); optimizer.step()               # backward replays graph B, its results reach .grad through autograd
#
# AutoGraph sits inside `forward` (GenS.forward, ImplicitSurface.forward): the first calls with a given signature run eagerly -- they are
# real steps, nothing has to be undone -- and record which outputs the caller's loss differentiates; then the forward is captured into one
# HIP graph and its backward (torch.autograd.grad of exactly those outputs) into a second one sharing its memory pool, the shape
# torch.cuda.make_graphed_callables has.  From then on a call copies the step's inputs into the static input tensors, puts the step's host
# random numbers into the page-locked buffer the graph's copy node reads, replays A and hands out the static outputs through an
# autograd.Function whose backward replays B.  A new signature (other shapes, another mode, parameters moved or frozen) is a new entry;
# whatever cannot be captured runs eagerly, in the same process, with a warning.
# ----------------------------------------------------------------------------------------------------------------------
import os
import warnings


# Capture mode of every graph in this file.  The default ("global") makes a runtime call from ANY thread illegal while a capture is open -- and
# torch.distributed's RCCL watchdog thread polls its events (hipEventQuery) whenever a collective is outstanding: a DDP-wrapped model that captured
# its step right after DDP's buffer broadcast was aborted from that thread now and then (tests/test_hip_ddp.py, one run in three).  "thread_local"
# restricts the checks to the capturing thread; kernels other threads (autograd's worker) enqueue on the capturing stream are captured either way.
_CAPTURE_MODE = os.environ.get("GENS_CAPTURE_MODE", "thread_local")


class _capturing:
    """`with torch.cuda.graph(...)` on a stream of its own, with a way back when the capture fails.  torch's context manager ends the capture in its
    __exit__ BEFORE it restores the current stream: a capture that was invalidated makes that call raise, and the thread stays on the side stream --
    which still reports "capturing", so the eager fall-through of AutoGraph ran into "operation not permitted when stream is capturing".  Here a failed
    capture puts the caller's stream back, drops the side stream for good (the next attempt gets a fresh one) and clears the runtime's sticky error."""

    side = {}          # device -> the stream captures run on (one for all captures, as torch's own default: the forward and backward graphs of an entry
                       # share a memory pool, whose blocks are reused only within the stream that freed them); replaced after a failed capture

    def __init__(self, graph, pool=None):
        self.prev = torch.cuda.current_stream()
        self.dev = torch.cuda.current_device()
        if self.dev not in _capturing.side:
            _capturing.side[self.dev] = torch.cuda.Stream()
        self.inner = torch.cuda.graph(graph, pool=pool, stream=_capturing.side[self.dev], capture_error_mode=_CAPTURE_MODE)

    def __enter__(self):
        return self.inner.__enter__()

    def __exit__(self, *exc):
        try:
            return self.inner.__exit__(*exc)
        except BaseException:
            torch.cuda.set_stream(self.prev)
            _capturing.side.pop(self.dev, None)
            _clear_sticky_hip_error()                                       # (the next launch check would raise it again)
            # capture_begin put the device's default generator into its capture mode and only capture_end's LAST step takes it out again: left there,
            # the next torch.randn on the device raises "Offset increment outside graph capture".  A clone of the state is a state outside any capture.
            try:
                gen = torch.cuda.default_generators[torch.cuda.current_device()]
                gen.graphsafe_set_state(gen.clone_state())
            except (AttributeError, RuntimeError):
                pass
            raise



_auto_off_depth = 0


class no_auto_graph:
    """Context: every GenS.forward / ImplicitSurface.forward call inside runs eagerly (no warm-up count, no capture, no replay) -- for a caller that
    captures the step itself (GraphedStep)."""

    def __enter__(self):
        global _auto_off_depth
        _auto_off_depth += 1
        return self

    def __exit__(self, *exc):
        global _auto_off_depth
        _auto_off_depth -= 1
        return False


def _loaded_hip_runtime():
    """Path of the libamdhip64 THIS process has mapped (torch's: possibly a versioned file bundled in torch/lib).  Opening "libamdhip64.so" by
    name could map a second runtime, whose hipGetLastError knows nothing of the first one's sticky error."""
    try:
        with open("/proc/self/maps") as f:
            for line in f:
                path = line.rstrip("\n").split(" ")[-1]
                if "libamdhip64.so" in os.path.basename(path):
                    return path
    except OSError:
        pass
    return None


def _clear_sticky_hip_error():
    path = _loaded_hip_runtime()
    if path is None:
        return
    try:
        import ctypes
        ctypes.CDLL(path).hipGetLastError()                    # already mapped: dlopen hands back the same handle; returns AND clears the error
    except OSError:
        pass


def auto_graph_enabled():
    """GENS_AUTO_GRAPH=0 switches the captured path off (every call eager, as in rounds 1 - 4); so does an enclosing `no_auto_graph()`."""
    return _auto_off_depth == 0 and os.environ.get("GENS_AUTO_GRAPH", "1") not in ("0", "off", "false", "no")


def _same(t):
    return t


def _phase(what):
    """GENS_AG_SYNC=1 (a debugging aid): wait for the device after every phase of a captured step and say so -- a GPU memory fault is reported
    asynchronously, the last phase printed before it is the one that faulted."""
    if os.environ.get("GENS_AG_SYNC"):
        import sys
        torch.cuda.synchronize()
        sys.stderr.write("      [auto-graph] %s\n" % what)
        sys.stderr.flush()


def _swapped_parameters(module, alias_of):
    """Context: the parameters of `module` that require a gradient are replaced by their aliases (torch's own functional-call machinery)."""
    import contextlib
    if module is None:
        return contextlib.nullcontext()
    from torch.nn.utils.stateless import _reparametrize_module
    return _reparametrize_module(module, {n: alias_of(p) for n, p in module.named_parameters() if p.requires_grad})


class _Entry:
    """One captured (forward, backward) pair and the static tensors around it."""
    __slots__ = ("calls", "used", "state", "fwd", "bwd", "static_in", "scalar_dev", "scalar_val", "refs", "grad_inputs", "grad_static", "out_names",
                 "out_static", "out_const", "out_diff", "diff_index", "bwd_used", "bwd_all", "recapture", "surfaces", "draws", "deferred", "fwd_done",
                 "tick", "why_eager", "out_order", "generation", "mask_words")

    def __init__(self):
        self.calls, self.used, self.state, self.tick, self.why_eager, self.recapture, self.generation = 0, set(), "warm", 0, None, False, 0


class _Replay(torch.autograd.Function):
    """forward: replay graph A, hand out the static outputs; backward: copy the cotangents into the static cotangent buffers, replay graph B, hand
    out the static input gradients.  The tensor arguments are the ORIGINAL inputs that require a gradient (parameters, and copied inputs that
    carry a graph of their own): autograd routes what backward returns to them."""

    @staticmethod
    def forward(ctx, entry, owner, *grad_inputs):
        entry.fwd.replay()
        _phase("forward graph replayed")
        entry.generation += 1
        ctx.entry, ctx.owner, ctx.generation = entry, owner, entry.generation
        outs = tuple(o.detach() for o in entry.out_static)
        ctx.mark_non_differentiable(*[o for o, d in zip(outs, entry.out_diff) if not d])
        ctx.set_materialize_grads(False)
        return outs

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, *grads):
        entry, owner = ctx.entry, ctx.owner
        if ctx.generation != entry.generation:
            # the saved activations of a captured step are the graph's STATIC buffers, shared by every call of the entry: a second forward has
            # overwritten them (gradient accumulation over micro-batches, two model(...) calls before one backward).  The reference's loop never
            # does this; silently wrong gradients are not an option
            raise RuntimeError("gens_amd.graph.AutoGraph: backward of a captured step whose forward has been replayed again since (its activations "
                               "live in static graph memory).  Call backward() before the next forward of the same signature, or switch the "
                               "captured path off for this loop: model.auto_graph = False / GENS_AUTO_GRAPH=0")
        owner._before_backward(entry)                          # the errors the reference raises inside forward: before any update
        plan = entry.bwd_used
        if any(g is not None and i not in plan.index for i, g in enumerate(grads)):
            # the caller's loss differentiates an output it did not use while this step warmed up: this step's backward takes the graph that
            # covers EVERY differentiable output (absent cotangents are zeros, autograd's own convention for custom Functions); the next call
            # captures the step again for the new set
            entry.used |= {entry.out_names[i] for i in entry.diff_index if grads[i] is not None}
            entry.recapture = True
            owner.stats["superset_backward"] += 1
            plan = entry.bwd_all
        dst, src = [], []
        for k, i in enumerate(plan.pattern):
            g, buf = grads[i], plan.grad_out[k]
            if g is None:
                if not plan.zero[k]:
                    buf.zero_()
                    plan.zero[k] = True
            else:
                if g.data_ptr() != buf.data_ptr():
                    if g.shape == buf.shape and g.dtype == buf.dtype:
                        dst.append(buf)
                        src.append(g)
                    else:
                        buf.copy_(g)
                plan.zero[k] = False
        if dst:
            torch._foreach_copy_(dst, src)                      # the loss's cotangents in one launch
        # a .grad that still aliases a static gradient buffer (a loop that zeroes gradients in place instead of dropping them, or accumulates over
        # several backward passes): give it memory of its own before the replay overwrites that buffer
        for t, s in zip(entry.grad_inputs, plan.grad_in):
            g = t.grad if t.is_leaf else None
            if g is not None and s is not None and g.data_ptr() == s.data_ptr():
                t.grad = g.clone()
        _phase("cotangents copied")
        if plan.graph is not None:
            plan.graph.replay()
        _phase("backward graph replayed")
        return (None, None, *[None if s is None else s.detach() for s in plan.grad_in])


class _Backward:
    """One captured backward: the output indices it differentiates, their static cotangent buffers, the graph, the static input gradients."""
    __slots__ = ("pattern", "index", "grad_out", "zero", "graph", "grad_in")


class AutoGraph:
    """Per-module cache of captured training forwards.  See the comment block above.

    run(key, copied, refs, scalars, body, surfaces):
      key      hashable signature of everything that shapes the launch sequence besides the tensors (mode, flags); shapes, dtypes, devices,
               requires_grad and -- for `refs` -- addresses are added here
      copied   {name: tensor} inputs that change from call to call: copied into static tensors before every replay
      refs     [tensor] inputs that persist between calls (parameters, buffers, leaf tensors the caller optimises): used where they are; when one
               moves, the signature changes
      scalars  {name: float} numbers the kernels read at launch time: kept in one-element device tensors the captured kernels read by address
      body     body(copied, scalars, alias) -> {name: tensor | anything}: the eager implementation; called with the caller's tensors and floats
               while warming up, with the static tensors and the device scalars under capture.  alias(t): what the body must use in place
               of a persistent tensor t it takes from `refs` itself (identity while warming up; under capture the alias leaf, see _capture)
      module   the nn.Module whose parameters the body reads through attribute access: swapped for their aliases during the capture
      surfaces the ImplicitSurface modules whose host-generator draws and deferred checks the body uses
    """

    def __init__(self, warmup=2, max_entries=3):
        self.warmup, self.max_entries = warmup, max_entries
        self.entries = {}
        self._tick = 0
        self.stats = {"eager": 0, "captured": 0, "replayed": 0, "superset_backward": 0, "evicted": 0}

    # -- signature -----------------------------------------------------------------------------------------------------
    @staticmethod
    def _sig(t):
        return (tuple(t.shape), t.dtype, t.device.index, bool(t.requires_grad))

    def _key(self, key, copied, refs, scalars):
        # (refs: address + requires_grad only -- a full-size GenS has ~700 parameters and buffers, and this runs on every call)
        return (key, tuple((n, self._sig(t)) for n, t in copied.items()), tuple([t.data_ptr() for t in refs]), tuple([t.requires_grad for t in refs]),
                tuple(scalars))

    def reset(self):
        """Forget every captured step (their memory pools go back to the allocator)."""
        self.entries.clear()

    # -- the call ------------------------------------------------------------------------------------------------------
    def run(self, key, copied, refs, scalars, body, surfaces, module=None):
        full_key = self._key(key, copied, refs, scalars)
        entry = self.entries.get(full_key)
        if entry is None:
            entry = self.entries[full_key] = _Entry()
            self._evict()
        self._tick += 1
        entry.tick = self._tick
        entry.calls += 1
        if entry.state == "eager":
            self.stats["eager"] += 1
            return body(copied, scalars, _same)
        if entry.state == "warm" and entry.calls <= self.warmup:
            self.stats["eager"] += 1
            return self._observed(entry, body(copied, scalars, _same))
        if entry.state == "warm" or entry.recapture:           # (recapture: the caller's loss has started to differentiate another set of outputs)
            entry.recapture = False
            entry.fwd = entry.bwd_used = entry.bwd_all = None
            try:
                self._capture(entry, copied, refs, scalars, body, surfaces, module)
            except Exception as e:  # noqa: BLE001   (whatever the capture trips over: this signature stays eager, in this process)
                entry.state, entry.why_eager = "eager", f"{type(e).__name__}: {e}"
                entry.fwd = entry.bwd_used = entry.bwd_all = None
                try:
                    torch.cuda.current_stream().synchronize()
                except RuntimeError:
                    pass
                warnings.warn(f"gens_amd.graph.AutoGraph: this step cannot be captured into a HIP graph ({entry.why_eager}); it runs eagerly",
                              RuntimeWarning, stacklevel=3)
                self.stats["eager"] += 1
                return body(copied, scalars, _same)
        return self._replay(entry, copied, scalars)

    def _evict(self):
        while len(self.entries) > self.max_entries:
            oldest = min(self.entries, key=lambda k: self.entries[k].tick)
            del self.entries[oldest]
            self.stats["evicted"] += 1

    def _observed(self, entry, out):
        """An eager warm-up step: note which outputs receive a gradient from the caller's loss (the captured backward differentiates those)."""
        for name, t in out.items():
            if torch.is_tensor(t) and t.requires_grad:
                # (the hook holds the SET, not the entry: whatever keeps an eager step's autograd graph alive must not keep captured graphs alive)
                t.register_hook(lambda g, name=name, used=entry.used: used.add(name))
        return out

    # -- capture -------------------------------------------------------------------------------------------------------
    def _capture(self, entry, copied, refs, scalars, body, surfaces, module=None):
        dev = next(iter(copied.values())).device if copied else refs[0].device
        for s in surfaces:
            s.check_deferred()
        torch.cuda.synchronize()
        static_in = {}
        mask_words = {}
        for n, t in copied.items():
            st = t.detach().clone()
            if t.requires_grad:
                st.requires_grad_(True)
            static_in[n] = st
            # a mask volume that arrives with its bit-packed copy (the volume build writes both: ops.volume_build) keeps one on its static twin, so the
            # captured step holds no packing launch; every replay refreshes the words with the tensor (_replay)
            hit = getattr(t, "_gens_bits", None)
            if hit is not None and hit[0] == t._version and hit[1].device == t.device:
                mask_words[n] = hit[1].clone()
                st._gens_bits = (st._version, mask_words[n])
        scalar_dev = {n: torch.full((1,), float(v), device=dev, dtype=torch.float32) for n, v in scalars.items()}
        saved_rng = torch.get_rng_state()
        for s in surfaces:                                     # page-locked buffers of this capture's own (see below)
            s.begin_capture()
        # Every persistent tensor that wants a gradient enters the captured autograd graph through an ALIAS leaf (same memory, created under the
        # capture stream).  The parameter's own AccumulateGrad node may be alive from an earlier eager step -- whatever still holds that step's
        # loss holds it -- and it belongs to the stream that step ran on (the default stream): the engine would then fork the capture to that
        # stream for every parameter, which ends this HIP runtime's hipStreamEndCapture in a segmentation fault (scripts/probe/auto_graph_probe*.py).
        aliases = {}

        def alias_of(t):
            if not (torch.is_tensor(t) and t.requires_grad):
                return t
            a = aliases.get(id(t))
            if a is None:
                a = aliases[id(t)] = t.detach().requires_grad_(True)
            return a
        fwd = torch.cuda.CUDAGraph()
        taken = []
        try:
            with _capturing(fwd):
                with torch.enable_grad(), _swapped_parameters(module, alias_of):
                    out = body(static_in, scalar_dev, alias_of)
        finally:
            # whether the capture went through or not: the capture pass drew one step's numbers without running a step (a failed capture falls
            # through to the EAGER step, which must see the generator where the caller left it), and the surfaces get buffers of their own back
            torch.set_rng_state(saved_rng)
            for s in surfaces:
                taken.append((s,) + tuple(s.end_capture()))
        # the page-locked buffers the captured copy nodes read from / write to now belong to this entry: an eager call of another signature
        # must not reallocate or refill them behind the graph's back
        entry.draws, entry.deferred = [], []
        for s, buf, layout, words in taken:
            entry.draws.append((s, buf, layout))
            entry.deferred.append((s, words))
        names, tensors, consts = [], [], {}
        entry.out_order = list(out)
        for n, v in out.items():
            if torch.is_tensor(v):
                names.append(n)
                tensors.append(v)
            else:
                consts[n] = v
        grad_inputs = [t for t in refs if t.requires_grad] + [t for t in copied.values() if t.requires_grad]
        grad_static = [alias_of(t) for t in refs if t.requires_grad] + [static_in[n] for n, t in copied.items() if t.requires_grad]
        diff = [bool(t.requires_grad) for t in tensors]
        used = entry.used if entry.used else {n for n, d in zip(names, diff) if d}      # (no backward seen while warming up: every differentiable output)
        pattern = tuple(i for i, (n, d) in enumerate(zip(names, diff)) if d and n in used)
        entry.fwd, entry.static_in, entry.scalar_dev, entry.scalar_val = fwd, static_in, scalar_dev, {n: float(v) for n, v in scalars.items()}
        entry.refs, entry.grad_inputs, entry.grad_static = list(refs), grad_inputs, grad_static
        entry.out_names, entry.out_static, entry.out_const, entry.out_diff = names, tensors, consts, diff
        entry.diff_index = [i for i, d in enumerate(diff) if d]
        entry.surfaces = list(surfaces)
        entry.mask_words = mask_words
        entry.fwd_done = None
        every = tuple(entry.diff_index)
        entry.bwd_used = self._capture_backward(entry, pattern)
        entry.bwd_all = entry.bwd_used if pattern == every else self._capture_backward(entry, every)
        entry.state = "captured"
        self.stats["captured"] += 1

    def _capture_backward(self, entry, pattern):
        plan = _Backward()
        plan.pattern, plan.index = tuple(pattern), frozenset(pattern)
        plan.grad_out = [torch.zeros_like(entry.out_static[i], memory_format=torch.contiguous_format) for i in pattern]
        plan.zero = [True] * len(pattern)
        plan.graph, plan.grad_in = None, [None] * len(entry.grad_static)
        if not pattern or not entry.grad_static:
            return plan
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with _capturing(graph, pool=entry.fwd.pool()):
            # (retain_graph: a second backward -- the one over every differentiable output -- is captured through the same autograd graph)
            grad_in = torch.autograd.grad([entry.out_static[i] for i in pattern], entry.grad_static, plan.grad_out, retain_graph=True, allow_unused=True)
        plan.graph, plan.grad_in = graph, list(grad_in)
        return plan

    # -- replay --------------------------------------------------------------------------------------------------------
    def _replay(self, entry, copied, scalars):
        for s in entry.surfaces:
            s.check_deferred()                                 # what the previous step left to verify, if its backward never ran
        dst, src, wdst, wsrc = [], [], [], []
        for n, t in copied.items():
            st = entry.static_in[n]
            if t.data_ptr() != st.data_ptr():
                dst.append(st.detach())
                src.append(t)
            words = entry.mask_words.get(n)
            if words is not None:
                hit = getattr(t, "_gens_bits", None)
                if hit is not None and hit[0] == t._version and hit[1].shape == words.shape and hit[1].device == words.device:
                    wdst.append(words)
                    wsrc.append(hit[1])
                else:                                          # (this step's mask came without its words: pack them now)
                    from . import lib as L
                    tc = t.detach().reshape(-1).contiguous()
                    L.call("gens_pack_mask_bits", L.ptr(tc), tc.numel(), L.ptr(words, torch.int32), L.stream())
        if dst:
            torch._foreach_copy_(dst, src, non_blocking=True)   # the step's ~ ten inputs in one or two launches (per dtype), not one each
        if wdst:
            torch._foreach_copy_(wdst, wsrc, non_blocking=True)
        for n, v in scalars.items():
            v = float(v)
            if v != entry.scalar_val[n]:
                entry.scalar_dev[n].fill_(v)
                entry.scalar_val[n] = v
        if entry.fwd_done is not None:
            entry.fwd_done.synchronize()                       # the previous replay's copy node has read the page-locked draws (free after a loss read-back)
        for s, buf, layout in entry.draws:
            if buf is not None:
                s.refresh_host_draws(buf, layout)              # the reference's generator, its order (implicit_surface.py:362, then :256)
        _phase("inputs copied, host draws refreshed")
        outs = _Replay.apply(entry, self, *entry.grad_inputs)
        entry.fwd_done = torch.cuda.Event()
        entry.fwd_done.record()
        for s, host in entry.deferred:                         # the step's device-side checks arrive in this entry's page-locked words
            if host is not None:
                s._deferred_host, s._deferred = host, entry.fwd_done
        self.stats["replayed"] += 1
        by_name = dict(zip(entry.out_names, outs))
        return {n: by_name[n] if n in by_name else entry.out_const[n] for n in entry.out_order}

    def _before_backward(self, entry):
        for s in entry.surfaces:
            s.check_deferred()                                 # waits for the forward replay, raises "No valid pseudo pts!" / a singular camera matrix

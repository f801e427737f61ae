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
        with torch.cuda.stream(side):                          # (warm-up off the default stream, as torch.cuda.graph requires)
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
        with torch.cuda.graph(self.graph):
            self.loss = body()
        torch.set_rng_state(saved_rng)                         # (the capture pass drew one step's numbers without running a step)
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

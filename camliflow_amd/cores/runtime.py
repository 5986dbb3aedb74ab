"""Which implementation the composite operators of the cores use.

``'hip'``       (default) every composite op with a fused gfx950 kernel calls it through the C-ABI;
                tensors must live on the GPU and the library must be present -- otherwise the call
                raises.  This is the product path.
``'composed'``  the same math composed from torch primitives exactly as the reference writes it
                (models/utils.py, models/point_conv.py, models/raft_core.py ...).  Device-agnostic;
                it is the fp32 torch reference the fused kernels are tested against, and what the
                CPU tests / the cpu_baseline leg of bench.py select explicitly.

A CPU tensor under 'hip' is an error, not a fallback.  A handful of call sites DO take the composed
formulation under 'hip' when a precondition of their kernel fails (autocast active, a shape outside the
kernel's range, an op variant without a kernel); every one of them reports through ``fallback()`` below,
so that
  * ``CAMLI_STRICT=1`` / ``set_strict(True)`` turns each such switch into an error, and
  * ``census()`` lists, per op, how many calls ran fused and how many composed (bench.py and the
    full-size tests print it, so it is known which ops of a configuration actually hit HIP).
"""
import collections
import contextlib
import os
import weakref

_BACKEND = 'hip'


def _stream_priority(lane):
    """HIP stream priority of a lane's stream (CAMLI_PRIO_SIDE | _AUX | _WGRAD; 0 = the default stream's, -1 = high: torch
    clamps to what the device offers and has nothing BELOW the default, so the point lane is put behind the image lane by
    raising the image lane's streams, see bench.py CAMLI_PRIO_MAIN).  Measured in round 6 and left at 0: a high-priority stream
    beside normal ones costs the two-lane step 65 ms on this stack (profiles/r06_experiments.txt 15c)."""
    return int(os.environ.get('CAMLI_PRIO_' + lane, '0'))


def backend():
    return _BACKEND


def set_backend(name):
    global _BACKEND
    if name not in ('hip', 'composed'):
        raise ValueError("backend must be 'hip' or 'composed', got %r" % (name,))
    _BACKEND = name


@contextlib.contextmanager
def use_backend(name):
    prev = backend()
    set_backend(name)
    try:
        yield
    finally:
        set_backend(prev)


def fused():
    return _BACKEND == 'hip'


# ------------------------------------------------------------------------------------------------
# strict mode + census of fused vs composed calls under the 'hip' backend
# ------------------------------------------------------------------------------------------------
_STRICT = os.environ.get('CAMLI_STRICT', '0') == '1'
_CENSUS = {'fused': collections.Counter(), 'composed': collections.Counter()}
_CENSUS_ON = os.environ.get('CAMLI_CENSUS', '0') == '1'


class CamliStrictError(RuntimeError):
    pass


def set_strict(enabled):
    global _STRICT
    _STRICT = bool(enabled)


def strict():
    return _STRICT


def set_census(enabled):
    global _CENSUS_ON
    _CENSUS_ON = bool(enabled)
    from ..csrc import _lib
    _lib._CENSUS = _CENSUS['fused'] if _CENSUS_ON else None


def census_on():
    return _CENSUS_ON


def reset_census():
    _CENSUS['fused'].clear()
    _CENSUS['composed'].clear()


def census():
    """{'fused': {entry point: launches}, 'composed': {'op: reason': calls}} since the last reset."""
    return {'fused': dict(_CENSUS['fused']), 'composed': dict(_CENSUS['composed'])}


def atomics_ok(op):
    """False under ``torch.use_deterministic_algorithms(True)`` for the fused ops whose adjoint (or reduction)
    accumulates with float atomics -- bias gradients, the masked-L2 sums, the
    interpolation / max-pool / up-sampling scatters.  Those ops then run the torch composition, which torch makes
    deterministic (or refuses loudly); the switch is recorded in the census and is not a strict-mode violation.
    The atomic-free kernels (sorted gather adjoints, PointConv mixing, the all-pairs and point cost-volume lookups,
    the weight network, the SK gate since round 3) stay on HIP: they are bit-reproducible by construction."""
    import torch
    if not torch.are_deterministic_algorithms_enabled():
        return True
    if _CENSUS_ON:
        _CENSUS['composed']['%s: deterministic algorithms requested (float atomics in the fused adjoint)' % op] += 1
    return False


def fallback(op, reason):
    """Called by a core op that has a fused kernel but is about to run the torch-composed formulation
    although the backend is 'hip'.  Raises in strict mode, otherwise records the event."""
    if _BACKEND != 'hip':
        return
    if _STRICT:
        raise CamliStrictError("%s: would run the torch-composed formulation under the 'hip' backend (%s); "
                               "CAMLI_STRICT=1 forbids it" % (op, reason))
    if _CENSUS_ON:
        _CENSUS['composed']['%s: %s' % (op, reason)] += 1


# ------------------------------------------------------------------------------------------------
# deferred parameter gradients: the 12 GRU iterations share their parameters, so autograd adds a
# fresh weight / bias gradient into .grad after every call (~1,200 small add + sum launches per
# step).  With this switch on, the fused 1x1-convolution and bias/activation nodes accumulate into
# one buffer per parameter inside their own kernels (GEMM with beta = 1, float atomics) and a
# callback at the end of backward() moves the totals into .grad.  Opt-in because it bypasses the
# functional API: torch.autograd.grad(...) would not see these gradients (training loops call
# .backward()).  Parameters that have hooks, and every parameter of a model wrapped in
# DistributedDataParallel, are excluded automatically (deferral_target).
# ------------------------------------------------------------------------------------------------
_DEFER_PARAM_GRADS = False

GEMM_TUNING_FILE = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'gemm_tuning_gfx950.csv')


def use_tuned_gemms(path=None):
    """Let PyTorch's TunableOp pick, per GEMM shape, the hipBLASLt / rocBLAS solution recorded in ``gemm_tuning_gfx950.csv``
    (533 shapes: the GEMMs of the four bench configurations, tuned on an MI355X with this image by running bench.py once under
    PYTORCH_TUNABLEOP_ENABLED=1).  The point MLPs, the 1x1 convolutions and their batched weight / data gradients go through
    ``at::cuda::blas``; the library heuristics' first choice is not the fastest for a third of those shapes: -3.5 ... -4.7 ms
    per training step (A/B on one box).  Tuning itself stays OFF (a shape that is not in the file runs the default solution),
    the file's validators (PyTorch / HIP / hipBLASLt / rocBLAS versions, gfx950) must match or TunableOp ignores it.
    Returns True when the table was installed.  CAMLI_TUNED_GEMMS=0 or an explicit PYTORCH_TUNABLEOP_ENABLED leave
    TunableOp alone."""
    import torch
    if os.environ.get('CAMLI_TUNED_GEMMS', '1') == '0' or 'PYTORCH_TUNABLEOP_ENABLED' in os.environ:
        return False
    path = path or GEMM_TUNING_FILE
    if not (torch.cuda.is_available() and os.path.exists(path)):
        return False
    try:
        tunable = torch.cuda.tunable
        tunable.enable(True)
        tunable.tuning_enable(False)
        tunable.set_filename(path, insert_device_ordinal=False)
        return bool(tunable.read_file(path))
    except Exception as exc:          # an older / differently built torch: run on the library defaults
        import warnings
        warnings.warn('camliflow_amd: tuned GEMM table not installed (%s)' % exc, RuntimeWarning)
        try:
            torch.cuda.tunable.enable(False)
        except Exception:
            pass
        return False


# Under torch.autocast the reference's update block runs its convolutions in the reduced precision (train.py:113,147-152).
# CAMLI_AUTOCAST_OWN=1: the update block's own fp32 kernels (GRU2D's tap convolutions, the Winograd 3x3s, their fused
# epilogues) also run inside an autocast region -- in fp32, i.e. at least the precision asked for -- instead of handing those
# layers to the library's reduced-precision kernels.
_AUTOCAST_OWN = os.environ.get('CAMLI_AUTOCAST_OWN', '0') == '1'


def own_kernels_allowed():
    """The fp32 own-kernel routes of the update block may be taken here: always outside autocast, inside it only when asked."""
    import torch
    return _AUTOCAST_OWN or not torch.is_autocast_enabled()


def set_deferred_param_grads(enabled):
    global _DEFER_PARAM_GRADS
    _DEFER_PARAM_GRADS = bool(enabled)


def deferred_param_grads():
    return _DEFER_PARAM_GRADS


def deferral_target(tensor):
    """The Parameter a fused node may accumulate into, or None.  Called in the node's FORWARD.

    Deferral bypasses the parameter's AccumulateGrad node, so everything hooked onto that node or onto the parameter
    would be skipped.  DistributedDataParallel is the case that matters: its reducer registers C++ post-hooks on the
    grad accumulators -- invisible from Python (``param._backward_hooks`` stays None) -- so DDP is recognised by the
    forward running inside ``DistributedDataParallel.forward`` (``_active_ddp_module``), and Python-level hooks by their
    dicts.  In all these cases the node returns its gradient to autograd in the ordinary way (correct, just not deferred)."""
    if not (_DEFER_PARAM_GRADS and tensor.is_leaf and tensor.requires_grad):
        return None
    PARAM_GRADS.drop_stale()        # accumulators of backwards that raised (no backward is running now)
    if getattr(tensor, '_backward_hooks', None) or getattr(tensor, '_post_accumulate_grad_hooks', None):
        return None
    try:
        from torch.nn.parallel import DistributedDataParallel as DDP
        if getattr(DDP, '_active_ddp_module', None) is not None:
            return None
    except ImportError:
        pass
    return tensor


class _ParamGradSink:
    """Per-parameter accumulators, one table per running backward (graph task): a re-entrant backward (activation
    checkpointing) has its own table and flush, and does not disturb the outer one."""
    MAX_TABLES = 4      # depth guard; tables of backwards that raised are dropped by the next forward (drop_stale)

    def __init__(self):
        self.tables = {}        # graph-task id -> {id(param): [param, accumulator, reduce_batch, stream]}

    @property
    def entries(self):
        return {k: v for table in self.tables.values() for k, v in table.items()}

    @property
    def armed(self):
        return bool(self.tables)

    def slot(self, param, make, reduce_batch):
        import torch
        task = torch._C._current_graph_task_id()
        table = self.tables.get(task)
        if table is None:
            table = self.tables[task] = {}
            torch.autograd.variable.Variable._execution_engine.queue_callback(lambda task=task: self.flush(task))
            while len(self.tables) > self.MAX_TABLES:       # leftovers of failed backwards inside one running backward
                del self.tables[min(self.tables)]
        entry = table.get(id(param))
        if entry is None:
            entry = table[id(param)] = [param, make(), reduce_batch, None]
        entry[3] = torch.cuda.current_stream(param.device) if param.is_cuda else None
        return entry[1]

    def drop_stale(self):
        """Called from the FORWARD of the deferring nodes (deferral_target).  A forward that runs while no backward is
        executing (graph-task id -1) finds tables only if earlier backwards raised -- their flush callbacks never ran and
        their accumulators ([B,Co,Ci] GEMM buffers, weight-sized tensors) would stay allocated for good, exactly in the
        out-of-memory recovery case.  A forward inside a backward (activation checkpointing) leaves everything alone."""
        if self.tables:
            import torch
            if torch._C._current_graph_task_id() == -1:
                self.tables.clear()

    def flush(self, task):
        import torch
        table = self.tables.pop(task, {})
        for stale in [t for t in self.tables if t > task]:     # nested backwards end before their parent: leftovers
            del self.tables[stale]
        with torch.no_grad():
            for param, acc, reduce_batch, stream in table.values():
                if stream is not None:
                    current = torch.cuda.current_stream(param.device)
                    if stream != current:
                        current.wait_stream(stream)          # the accumulator was last written on another lane
                        acc.record_stream(current)
                grad = (acc.sum(0) if reduce_batch else acc).view_as(param)
                param.grad = grad if param.grad is None else param.grad + grad


PARAM_GRADS = _ParamGradSink()


# ------------------------------------------------------------------------------------------------
# two-lane execution: point branch on a side HIP stream next to the image branch
# ------------------------------------------------------------------------------------------------
_OVERLAP = False
_PRIME_FIRST_PASS = True
_PRIMED = set()
_side_streams = {}


def streamk_safe():
    """Two GEMM-issuing streams are only safe when hipBLASLt's stream-K kernels run data-parallel: two stream-K GEMMs
    of one handle on two streams spin on each other's flags for good (found in round 3).  True when
    TENSILE_STREAMK_DATA_PARALLEL=1 is in the environment AND it got there before this process created its CUDA
    context (preset by the user / launcher, or set by the package import ahead of the first GPU call)."""
    import camliflow_amd
    return os.environ.get('TENSILE_STREAMK_DATA_PARALLEL') == '1' and camliflow_amd.STREAMK_SET_BEFORE_CUDA


def set_overlap(enabled, prime_first_pass=True):
    """Run the 3-D (point) branch of the fused model on a second HIP stream (and, inside such a pass, the independent
    chains of the image lane on auxiliary streams, ``Branch``).  The point kernels are small (B*2048 points) and leave
    most CUs idle; the image branch's convolutions do not depend on them between fusion points, so the lanes overlap.
    Results are unchanged.

    REFUSED (one lane, with a RuntimeWarning) unless ``streamk_safe()``: the lanes both issue library GEMMs, and
    hipBLASLt's stream-K GEMMs of one handle on two streams dead-lock the GPU (100 % busy for good).  The package import
    sets TENSILE_STREAMK_DATA_PARALLEL=1, but that only helps before the first GEMM of the process: a host program that
    initialised CUDA first, or a user who set the variable to 0, gets one lane.  CAMLI_OVERLAP_FORCE=1 overrides (for
    a caller that knows no stream-K solution is in play).

    ``prime_first_pass`` (default): the FIRST pass of a process for a given (device, input signature) still runs on
    one stream -- forward and, because autograd replays a node on the stream of its forward, its backward.  The first
    step of a process is where the libraries do their first-use work (MIOpen / hipBLASLt solution look-ups, code-object
    loads): 20-60 s on most boxes of this pool, 231 s measured on a slow one, on ONE stream
    (profiles/r03_first_step_probe.txt).  Round 2 read that stall as a two-stream dead-lock; the real multi-stream
    dead-lock (stream-K, above) was found and fixed later in round 3.  Keeping that step on one stream costs nothing (it
    is never a timed step) and keeps a first-use problem from being confused with a stream-ordering problem again.  It
    lives here, not in a benchmark script, so that EVERY caller of ``set_overlap(True)`` gets it."""
    global _OVERLAP, _PRIME_FIRST_PASS, _LANES_LIVE
    enabled = bool(enabled)
    if enabled and not streamk_safe() and os.environ.get('CAMLI_OVERLAP_FORCE') != '1':
        import warnings
        warnings.warn('camliflow_amd: multi-stream execution refused -- TENSILE_STREAMK_DATA_PARALLEL=1 was not in the '
                      'environment before the CUDA context of this process existed (value now: %r); hipBLASLt stream-K '
                      'GEMMs on two streams dead-lock the GPU.  Import camliflow_amd (or export the variable) before the '
                      'first GPU call.  Running on one stream.' % os.environ.get('TENSILE_STREAMK_DATA_PARALLEL'),
                      RuntimeWarning, stacklevel=2)
        enabled = False
    _OVERLAP = enabled
    _PRIME_FIRST_PASS = bool(prime_first_pass)
    if not _OVERLAP:
        _LANES_LIVE = False     # a model without Lanes of its own must not inherit the auxiliary streams of an earlier pass


def overlap():
    return _OVERLAP


def reset_lane_priming():
    _PRIMED.clear()


def _flatten(items):
    for it in items:
        if isinstance(it, (list, tuple)):
            yield from _flatten(it)
        elif it is not None:
            yield it


class Lanes:
    """Fork/join helper.  ``side()`` is a context that issues work on the side stream;
    ``to_side(...)`` / ``to_main(...)`` order the two streams at a hand-over and tell the caching
    allocator that the handed-over tensors are in use on the other stream."""

    def __init__(self, device, key=None):
        import torch
        self.enabled = _OVERLAP and _BACKEND == 'hip' and device.type == 'cuda'
        if self.enabled and _PRIME_FIRST_PASS:
            signature = (device.index, torch.is_grad_enabled(), key)
            if signature not in _PRIMED:        # first pass of this shape: one lane (see set_overlap)
                _PRIMED.add(signature)
                self.enabled = False
        global _LANES_LIVE, _LANES_OWNER
        _LANES_LIVE = self.enabled and os.environ.get('CAMLI_BRANCHES', '1') == '1'
        _LANES_OWNER = weakref.ref(self)    # the flag lives as long as the pass that set it (lanes_live)
        if self.enabled:
            self._torch = torch
            self.main = torch.cuda.current_stream(device)
            key = (device.index, self.main.cuda_stream)
            if key not in _side_streams:
                _side_streams[key] = torch.cuda.Stream(device, priority=_stream_priority('SIDE'))
            self.side_stream = _side_streams[key]

    def side(self):
        if not self.enabled:
            return contextlib.nullcontext()
        return self._torch.cuda.stream(self.side_stream)

    def to_side(self, *tensors):
        if self.enabled:
            self.side_stream.wait_stream(self.main)
            for t in _flatten(tensors):
                t.record_stream(self.side_stream)

    def to_main(self, *tensors):
        if self.enabled:
            self.main.wait_stream(self.side_stream)
            for t in _flatten(tensors):
                t.record_stream(self.main)

_LANES_LIVE = False     # the Lanes of the pass that is being issued are enabled (set by Lanes.__init__)
_LANES_OWNER = None     # weak reference to the Lanes object that set it


def lanes_live():
    """The auxiliary streams belong to the pass whose ``Lanes`` enabled them: once that object is gone (its forward has
    returned), a model that builds no Lanes of its own (CamLiPWC, the image-only RAFT, CamLiRAFT-L) does not inherit the
    flag of an earlier CamLiRAFT pass, even with the overlap left on."""
    return bool(_LANES_LIVE and (_LANES_OWNER is None or _LANES_OWNER() is not None))


_aux_streams = {}
_BRANCH_MASK = int(os.environ.get('CAMLI_BRANCH_MASK', '15'))      # bit = slot: 1 motion encoder, 2 mask head, 4 CLFM, 8 context encoder
_BRANCH_SHARE = os.environ.get('CAMLI_BRANCH_SHARE', '0') == '1'   # all slots on ONE auxiliary stream


class Branch:
    """A short independent chain of the IMAGE lane issued on an auxiliary stream (round 3): the 7x7 / 3x3 flow branch of
    MotionEncoder2D next to the correlation lookup + its 1x1 / 3x3 branch, the mask head next to the flow head.  The
    chains are independent until they are concatenated / consumed, one of them is a convolution that cannot fill the
    chip (2 -> 128 channels) or a run of HBM-bound epilogues, the other one is bound by the matrix cores -- on one
    stream they run one after the other.  autograd replays every node on the stream of its forward, so the adjoints
    overlap the same way.  Usage:

        br = runtime.Branch(x)            # fork here: the aux stream waits for what the current stream holds NOW
        with br:
            y = chain(x)                  # issued on the aux stream
        ...                               # other work on the current stream
        br.join(y)                        # the current stream waits for the chain; y may be used on it

    Enabled only inside a pass whose Lanes are enabled (two-lane execution, not the priming pass), so the one-lane
    configurations stay exactly one stream."""

    def __init__(self, *inputs, slot=0):
        import torch
        first = next((t for t in _flatten(inputs) if t is not None), None)
        # not under HIP-graph capture.  Round 4 bisected it per slot on --config kitti (CAMLI_BRANCH_IN_CAPTURE=1 lifts this
        # guard, CAMLI_BRANCH_MASK picks the slots): the motion-encoder, mask-head and context-encoder branches capture and
        # replay correctly but make the replayed batch-1 step SLOWER (143.7 / 144.2 / 150.1 ms against 142.1 ms without:
        # a replayed graph is not enqueue-bound and these chains are too short to pay for their fork / join at batch 1);
        # with the CLFM direction forked in, hipStreamEndCapture itself crashes (SIGSEGV inside torch's capture_end where
        # an un-joinable fork should come back as hipErrorStreamCaptureUnjoined) -- profiles/r04_branch_capture_bisect.txt.
        # So the replayed configurations keep exactly their two lanes.
        self.enabled = bool(lanes_live() and _OVERLAP and _BACKEND == 'hip' and first is not None and first.is_cuda
                            and (_BRANCH_MASK >> slot) & 1
                            and (os.environ.get('CAMLI_BRANCH_IN_CAPTURE') == '1' or not torch.cuda.is_current_stream_capturing()))
        if self.enabled:
            self._torch = torch
            dev = first.device
            self.main = torch.cuda.current_stream(dev)
            key = (dev.index, self.main.cuda_stream, 0 if _BRANCH_SHARE else slot)
            if key not in _aux_streams:
                _aux_streams[key] = torch.cuda.Stream(dev, priority=_stream_priority('AUX'))
            self.aux = _aux_streams[key]
            self.aux.wait_stream(self.main)
            for t in _flatten(inputs):
                if t is not None:
                    t.record_stream(self.aux)

    def __enter__(self):
        if self.enabled:
            self._ctx = self._torch.cuda.stream(self.aux)
            self._ctx.__enter__()
        return self

    def __exit__(self, *exc):
        if self.enabled:
            return self._ctx.__exit__(*exc)
        return False

    def join(self, *outputs):
        if self.enabled:
            current = self._torch.cuda.current_stream(self.aux.device)
            current.wait_stream(self.aux)
            for t in _flatten(outputs):
                if t is not None:
                    t.record_stream(current)


# ------------------------------------------------------------------------------------------------
# weight gradients beside the data-gradient chain (round 5)
# ------------------------------------------------------------------------------------------------
# In a backward pass the DATA gradient of a convolution is on the critical chain (the next adjoint needs it); its WEIGHT gradient is
# needed by nobody until the optimizer runs.  On one stream they run one after the other.  `wgrad_side(device)` hands out ONE
# auxiliary stream per (device, current stream) for weight gradients: the node makes it wait for what the current stream holds
# (`fork`), issues the weight gradient there -- accumulating into a per-pass total that only that stream touches -- and whoever
# hands the totals to autograd (`GRU2DPass`'s hub node, the deferred-parameter sink's flush) joins it first.  Enabled with the
# two-lane execution (never under graph capture); CAMLI_WGRAD_ASIDE=0 keeps everything on the current stream.
_wgrad_streams = {}
_WGRAD_ASIDE = os.environ.get('CAMLI_WGRAD_ASIDE', '1') != '0'


class _WgradSide:
    def __init__(self, main, side):
        self.main, self.side = main, side

    def fork(self, *tensors):
        """the side stream waits for everything the current stream holds now; `tensors` will be read / written there"""
        self.side.wait_stream(self.main)
        for t in _flatten(tensors):
            if t is not None:
                t.record_stream(self.side)

    def stream(self):
        import torch
        return torch.cuda.stream(self.side)

    def join(self):
        """the current stream waits for every weight gradient issued so far"""
        import torch
        torch.cuda.current_stream(self.side.device).wait_stream(self.side)


def wgrad_side(device):
    """The weight-gradient side stream of the current stream on `device`, or None where it does not apply.  Callers ask for
    it only in the backward of a pass whose FORWARD ran two-lane (they record ``lanes_live()`` at forward time -- in the
    backward the pass's Lanes object is gone): the priming pass and the models that build no Lanes (RAFTCore, CamLiRAFT-L,
    CamLiPWC) stay on one stream in both directions, as Branch promises."""
    import torch
    if not (_WGRAD_ASIDE and _OVERLAP and _BACKEND == 'hip' and device.type == 'cuda') or torch.cuda.is_current_stream_capturing():
        return None
    main = torch.cuda.current_stream(device)
    key = (device.index, main.cuda_stream)
    side = _wgrad_streams.get(key)
    if side is None:
        side = _wgrad_streams[key] = torch.cuda.Stream(device, priority=_stream_priority('WGRAD'))
    return _WgradSide(main, side)


if _CENSUS_ON:
    set_census(True)

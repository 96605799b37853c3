"""Launch-plan containers of the stage-2 denoiser's host side: device buffers planned once per geometry (Buf), strided
f16 row views into them (Rows: every activation is channels-last rows X[(b f y x)][C], so the reference's einops
permutes — resnet.py:14-16, attention.py:322-361, motion_module.py:157-180 — and torch.cat of the skip connections —
unet_blocks.py:644,754 — are index arithmetic), and the ordered list of librcdm_hip.so launches (Plan) that one
hipGraph capture replays per denoising step (RCDMs_pipeline.py:480-503)."""
import torch


class Buf:
    """A device buffer whose size is the max over all requests made while planning."""
    __slots__ = ("name", "nbytes", "t", "f16")

    def __init__(self, name, nbytes):
        self.name, self.nbytes, self.t = name, int(nbytes), None
        self.f16 = False    # holds f16 activation rows (Plan.rows): what numerics.numerics_report scans for range

    @property
    def ptr(self):
        return self.t.data_ptr()


class Rows:
    """View of f16 rows [M][C] with row stride ld (elements) inside a Buf at element offset off."""
    __slots__ = ("buf", "off", "M", "C", "ld")

    def __init__(self, buf, off, M, C, ld):
        self.buf, self.off, self.M, self.C, self.ld = buf, int(off), int(M), int(C), int(ld)

    @property
    def ptr(self):
        return self.buf.t.data_ptr() + 2 * self.off

    def ptr_key(self):
        """Identity of the first element (valid before the buffers are materialised, unlike .ptr)."""
        return (id(self.buf), self.off, self.ld)

    def cols(self, c0, c):
        return Rows(self.buf, self.off + c0, self.M, c, self.ld)

    def rows(self, r0, n):
        return Rows(self.buf, self.off + r0 * self.ld, n, self.C, self.ld)


class Plan:
    """Ordered launch list + the buffers it touches."""

    def __init__(self, device):
        self.device = torch.device(device)
        self.bufs = {}
        self.ops = []
        self.tags = []  # one label per op (kind + shape): tools/opprof.py aggregates per-op timings by it
        self.keep = []  # tensors that must outlive the plan (packed weights etc.)
        self.n_launch = 0
        self.op_weights = {}   # op index -> weight tensor of a GEMM / conv op (tools/prefetch_bound.py)
        self.op_desc = {}      # op index -> ("gemm", GemmDesc, producer, consumer) | ("conv", ConvDesc)   [tools/ceiling.py]
        self.lnx_sites = []    # deferred-LayerNorm consumers: (op index, A rows, colsum S, tag)          [numerics_report]
        self.attn_sites = []   # self-attention sites: (op index, tag, weight-norm score bound, wide flag, q rows, k rows, heads, d)

    def scratch(self, name, nbytes):
        b = self.bufs.get(name)
        if b is None:
            b = self.bufs[name] = Buf(name, nbytes)
        elif nbytes > b.nbytes:
            assert b.t is None, "scratch grown after materialize"
            b.nbytes = int(nbytes)
        return b

    def new(self, name, nbytes):
        assert name not in self.bufs, name
        b = self.bufs[name] = Buf(name, nbytes)
        return b

    def rows(self, name, M, C, ld=None, unique=False):
        ld = ld or C
        buf = (self.new if unique else self.scratch)(name, M * ld * 2)
        buf.f16 = True
        return Rows(buf, 0, M, C, ld)

    def materialize(self):
        for b in self.bufs.values():
            if b.t is None:
                b.t = torch.zeros(max(b.nbytes, 256), dtype=torch.uint8, device=self.device)

    def total_bytes(self):
        return sum(b.nbytes for b in self.bufs.values())

    def add(self, fn, tag="misc"):
        self.ops.append(fn)
        self.tags.append(tag)

    def run(self, ops=None):
        for op in (self.ops if ops is None else ops):
            op()


class _NS:
    def __init__(self, **kw):
        self.__dict__.update(kw)


class Geo:
    """b samples x f frames of H x W latent pixels."""

    def __init__(self, b, f, H, W):
        self.b, self.f, self.H, self.W = b, f, H, W
        self.n_img = b * f
        self.hw = H * W
        self.M = self.n_img * self.hw

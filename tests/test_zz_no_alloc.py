"""tp_run_lines_device_noalloc against tp_run_lines_device: same inputs, every output bit for bit, on two calls in a
row (the second with a smaller input, so what the first left in the static working memory must not show), with the
detail arrays, with every file in the general kernel, and the refusals at the limits."""

import numpy as np
import pytest

from helpers import Config

pytestmark = pytest.mark.gpu


class DeviceRun:
    """Inputs of one corpus and one set of output buffers in HBM."""

    def __init__(self, pipe, corpus, details=False):
        import torch
        from imagecaptioner_amd._abi import tp_device_io
        self.torch = torch
        dev = torch.device("cuda", 0)
        n = corpus.n_files
        self.n = n
        self.arena = torch.cat([torch.from_numpy(corpus.arena).to(dev), torch.zeros(64, dtype=torch.uint8, device=dev)])
        self.offsets = torch.from_numpy(corpus.offsets.astype(np.int64)).to(dev)
        self.capacity = max(8 * int(corpus.offsets[-1]), 1 << 16)
        self.caps = torch.full((self.capacity,), 0xEE, dtype=torch.uint8, device=dev)
        self.cap_off = torch.full((n + 1,), -1, dtype=torch.int64, device=dev)
        self.ver = torch.full((max(n, 1) * 16,), 0xEE, dtype=torch.uint8, device=dev)
        self.hist = torch.zeros(pipe.info.n_strings + 1, dtype=torch.int64, device=dev)
        self.ctr = torch.zeros(8, dtype=torch.int64, device=dev)
        io = tp_device_io()
        io.arena, io.offsets, io.n_files = self.arena.data_ptr(), self.offsets.data_ptr(), n
        io.captions, io.caption_capacity, io.caption_offsets = self.caps.data_ptr(), self.capacity, self.cap_off.data_ptr()
        io.verdicts, io.histogram, io.counters = self.ver.data_ptr(), self.hist.data_ptr(), self.ctr.data_ptr()
        self.fired = self.facets = None
        if details:
            self.fired = torch.full((max(n * pipe.info.fired_words, 1),), -1, dtype=torch.int32, device=dev)
            self.facets = torch.full((max(n * pipe.info.n_facets, 1),), -1, dtype=torch.int16, device=dev)
            io.fired, io.facet_choice = self.fired.data_ptr(), self.facets.data_ptr()
        io.stream = torch.cuda.current_stream().cuda_stream
        self.io = io
        self.pipe = pipe
        self.fw, self.nf = pipe.info.fired_words, pipe.info.n_facets

    def run(self, noalloc):
        self.nbytes = self.pipe.run_lines_device(self.io, noalloc=noalloc)
        self.torch.cuda.synchronize()
        return self

    def outputs(self):
        out = {"nbytes": self.nbytes, "captions": self.caps[:self.nbytes].cpu().numpy(), "offsets": self.cap_off.cpu().numpy(),
               "verdicts": self.ver[:self.n * 16].cpu().numpy(), "hist": self.hist.cpu().numpy(), "ctr": self.ctr.cpu().numpy()}
        if self.fired is not None:
            out["fired"] = self.fired[:self.n * self.fw].cpu().numpy()
            out["facets"] = self.facets[:self.n * self.nf].cpu().numpy()
        return out


def same(a, b):
    assert a.keys() == b.keys()
    for k in a:
        assert np.array_equal(a[k], b[k]), k


def test_noalloc_matches_allocating_entry_point_twice_in_a_row():
    import torch
    from imagecaptioner_amd import TagPipe
    from imagecaptioner_amd.corpus import make_corpus
    big = make_corpus("c2", 6000, seed=77)
    p = TagPipe(Config(**big.config_kwargs))
    small = make_corpus("c2", 1300, seed=78)
    # the calls without allocation come first: nothing the other entry point grew can be what they work in
    a1 = DeviceRun(p, big).run(noalloc=True)
    free_before = torch.cuda.mem_get_info()[0]
    a2 = DeviceRun(p, small)
    a2_free = torch.cuda.mem_get_info()[0]
    a2.run(noalloc=True)
    assert torch.cuda.mem_get_info()[0] == a2_free <= free_before      # the call took no device memory
    assert p.last_timing()["total_ms"] == 0.0
    b1 = DeviceRun(p, big).run(noalloc=False)
    b2 = DeviceRun(p, small).run(noalloc=False)
    same(a1.outputs(), b1.outputs())
    same(a2.outputs(), b2.outputs())
    assert int(a1.ctr[0]) == big.n_files and int(a2.ctr[0]) == small.n_files
    # and once more after the allocating calls, the smaller input again
    same(DeviceRun(p, small).run(noalloc=True).outputs(), b2.outputs())


def test_noalloc_with_detail_arrays_and_other_tables():
    from imagecaptioner_amd import TagPipe
    from imagecaptioner_amd.corpus import make_corpus
    for name, n1, n2 in (("c5", 1500, 400), ("c3", 3000, 10), ("c1", 2000, 1)):
        c1, c2 = make_corpus(name, n1, seed=5), make_corpus(name, n2, seed=6)
        p = TagPipe(Config(**c1.config_kwargs))
        for c in (c1, c2):
            same(DeviceRun(p, c, details=True).run(noalloc=True).outputs(), DeviceRun(p, c, details=True).run(noalloc=False).outputs())


def test_noalloc_general_kernel_works_in_the_fixed_workspace(monkeypatch):
    """Every file through the thread-per-file kernel: its workspace is what is left of the static array, and a batch that
    does not fit is worked through in halves."""
    from imagecaptioner_amd import TagPipe
    from imagecaptioner_amd.corpus import make_corpus
    c1, c2 = make_corpus("c2", 5000, seed=9), make_corpus("c2", 700, seed=10)
    monkeypatch.setenv("TAGPIPE_NO_FAST", "1")
    p = TagPipe(Config(**c1.config_kwargs))
    monkeypatch.delenv("TAGPIPE_NO_FAST")
    for c in (c1, c2):
        a = DeviceRun(p, c).run(noalloc=True)
        assert p.last_timing()["general_files"] == c.n_files
        same(a.outputs(), DeviceRun(p, c).run(noalloc=False).outputs())


def test_noalloc_limits():
    from imagecaptioner_amd import TagPipe, TagPipeError
    from imagecaptioner_amd._abi import TP_E_CAPACITY, TP_E_INVALID_ARG
    from imagecaptioner_amd.corpus import make_corpus
    from imagecaptioner_amd.tagpipe import NOALLOC_MAX_FILES
    c = make_corpus("c1", NOALLOC_MAX_FILES + 1, seed=3)
    p = TagPipe(Config(**c.config_kwargs))
    r = DeviceRun(p, c)
    with pytest.raises(TagPipeError) as e:
        r.run(noalloc=True)
    assert e.value.code == TP_E_INVALID_ARG and str(NOALLOC_MAX_FILES) in str(e.value)
    # the largest batch it takes
    full = make_corpus("c1", NOALLOC_MAX_FILES, seed=3)
    same(DeviceRun(p, full).run(noalloc=True).outputs(), DeviceRun(p, full).run(noalloc=False).outputs())
    # a caption buffer that is too small is reported like the other entry point reports it
    r = DeviceRun(p, make_corpus("c1", 100, seed=4))
    want = r.run(noalloc=False).nbytes
    r.io.caption_capacity = want - 1
    with pytest.raises(TagPipeError) as e:
        r.run(noalloc=True)
    assert e.value.code == TP_E_CAPACITY and e.value.required == want
    # no files at all
    import types
    empty = types.SimpleNamespace(n_files=0, arena=np.zeros(0, np.uint8), offsets=np.zeros(1, np.uint64))
    r = DeviceRun(p, empty).run(noalloc=True)
    assert r.nbytes == 0 and int(r.cap_off[0]) == 0 and int(r.ctr[0]) == 0

"""The C-ABI library loads without a GPU and exports every function include/tagpipe.h declares; without a device
tp_create fails loudly instead of falling back to anything."""
import ctypes as C
import os
import re

import pytest

from helpers import ROOT, shipped_config
import imagecaptioner_amd
from imagecaptioner_amd import _abi, tagpipe


def header_functions():
    text = open(os.path.join(ROOT, "include", "tagpipe.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(tp_[a-z_0-9]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from imagecaptioner_amd import build
    build.build()
    lib = tagpipe.load_library()
    names = header_functions()
    assert set(names) == set(tagpipe.EXPORTS), (names, tagpipe.EXPORTS)
    for n in names:
        assert hasattr(lib, n), n


def test_struct_layouts_match_header():
    assert C.sizeof(_abi.tp_verdict) == 16
    assert C.sizeof(_abi.tp_counters) == 64
    assert C.sizeof(_abi.tp_source) == 24
    assert C.sizeof(_abi.tp_timing) == 56
    assert C.sizeof(_abi.tp_device_io) == 12 * 8
    # compile a probe with the real header and compare sizes / a few offsets
    import subprocess, tempfile, textwrap
    src = textwrap.dedent("""
        #include <stdio.h>
        #include <stddef.h>
        #include "tagpipe.h"
        int main(void) {
            printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu\\n", sizeof(tp_config), sizeof(tp_result), sizeof(tp_table_info), offsetof(tp_config, trigger),
                   offsetof(tp_config, line_weight), offsetof(tp_result, histogram), offsetof(tp_result, opaque), sizeof(tp_timing), offsetof(tp_timing, fused_reruns));
            return 0;
        }""")
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "p.c"), "w").write(src)
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), "-o", os.path.join(d, "p"), os.path.join(d, "p.c")])
        out = subprocess.check_output([os.path.join(d, "p")]).decode().split()
    got = [C.sizeof(_abi.tp_config), C.sizeof(_abi.tp_result), C.sizeof(_abi.tp_table_info), _abi.tp_config.trigger.offset,
           _abi.tp_config.line_weight.offset, _abi.tp_result.histogram.offset, _abi.tp_result.opaque.offset, C.sizeof(_abi.tp_timing),
           _abi.tp_timing.fused_reruns.offset]
    assert [int(x) for x in out] == got


def test_no_device_fails_loudly():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(imagecaptioner_amd.TagPipeError) as e:
        imagecaptioner_amd.TagPipe(shipped_config())
    assert "GPU" in str(e.value) or "HIP" in str(e.value)


def test_product_package_never_touches_the_oracle():
    pkg = os.path.join(ROOT, "imagecaptioner_amd")
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".h", ".hpp", ".cpp", ".hip", ".inc")):
                text = open(os.path.join(dirpath, fn), encoding="utf-8", errors="replace").read()
                assert "oracle" not in text.lower(), os.path.join(dirpath, fn)

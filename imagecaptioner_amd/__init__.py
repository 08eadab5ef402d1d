"""MI355X batch tag pipeline: HIP kernels behind a C ABI, with a thin ctypes host binding.

    from imagecaptioner_amd import Config, TagPipe
    pipe = TagPipe(Config(lexicon="rules/lexicon.json", vocabulary="top_tags.txt", minimal_subject_tags="1girl"))
    result = pipe.run_texts(["1girl, solo, blue_hair"])

The pipeline itself runs only on the GPU (imagecaptioner_amd/csrc); importing this
package does not need one, constructing a TagPipe does.
"""
from ._abi import Config  # noqa: F401
from .tagpipe import BatchResult, TagPipe, TagPipeError, load_library  # noqa: F401

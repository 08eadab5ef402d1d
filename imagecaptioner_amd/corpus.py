"""Seeded synthetic tag corpora and table files for the benchmark configurations.

Everything is derived from splitmix64 streams with stated seeds, so a corpus is a
pure function of (config name, n_files, seed).  Text is ASCII: comma-separated
tag lines, one per file, packed into a byte arena plus an offsets array — the
shape the reference's ParseTagCandidatesFromLine reads (Program.cs:574-587).

Configurations (names follow BASELINE.json's list):
  c1  shipped vocabulary / rules / profile, ~30 tags per file
  c2  20k-entry generated vocabulary + 20k-entry lexicon, ~40 tags per file
  c3  100k-entry lexicon with alias chains (depth 2-8) and 2-cycles
  c4  c2 tables with Zipf(1.1) tag choice (hot-tag histogram stress)
  c5  heavy-tailed tag counts up to 500, 64 action rules x 4 subtypes, 32 facets x 4 rules
"""
import json
import os

import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)
REFERENCE_DATA = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "reference_data")


def splitmix64(seed, n, stream=0, start=0):
    """n 64-bit values of the splitmix64 sequence started at `seed` (stream offsets the seed), from element `start` on:
    element i depends on (seed, stream, i) alone, so a slice of a sequence can be drawn without the elements before it."""
    with np.errstate(over="ignore"):
        z = (np.uint64(seed) + np.uint64(stream) * np.uint64(0xD1342543DE82EF95)
             + (np.arange(start + 1, start + n + 1, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15)))
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def _uniform01(bits):
    return (bits >> np.uint64(11)).astype(np.float64) * (1.0 / (1 << 53))


_COLORS = "red blue green black white pink purple yellow orange brown grey silver blonde aqua dark light".split()
_FACE = "hair eyes eyebrows bangs twintails ponytail braid ahoge blush smile frown tears".split()
_OUTFIT = "dress skirt shirt jacket coat hoodie sweater uniform kimono swimsuit socks boots gloves hat ribbon bow necktie blazer pants shorts".split()
_POSE = "standing sitting lying kneeling walking running jumping leaning holding arms_up hands_on_hips from_side from_behind".split()
_BACK = "indoors outdoors bedroom street city classroom park forest beach sky night sunset window chair office kitchen".split()
_STYLE = "lighting backlighting shadow bloom bokeh monochrome highres absurdres masterpiece detailed".split()
_MISC = "solo cat dog sword flower book cup star moon heart wings tail horns halo scarf bag umbrella guitar phone food cake tea".split()
_SYL = "ka ri no mi sa to ha ru yu ki a o e na ma ko ta shi zu ne".split()


def make_vocabulary(n, seed=1):
    """n distinct valid ASCII tags covering every ClassifyTag bucket (Program.cs:1103-1111)."""
    out, seen = [], set()

    def add(t):
        if t not in seen and len(t) <= 59:
            seen.add(t)
            out.append(t)

    for t in ("1girl", "solo", "long_hair", "smile", "looking_at_viewer", "open_mouth", "school_uniform", "barefoot", "toes", "feet", "penis", "hand", "oral"):
        add(t)
    groups = [_FACE, _OUTFIT, _POSE, _BACK, _STYLE, _MISC]
    for g in groups:
        for w in g:
            add(w)
    for c in _COLORS:
        for g in (_FACE, _OUTFIT, _MISC):
            for w in g:
                add(c + "_" + w)
    r = splitmix64(seed, 8 * n + 64, stream=7)
    k = 0
    while len(out) < n:
        a, b, c, d = (int(r[k]), int(r[k + 1]), int(r[k + 2]), int(r[k + 3]))
        k += 4
        if k + 4 >= len(r):
            r = splitmix64(seed + len(out), 8 * n + 64, stream=9)
            k = 0
        kind = a % 5
        if kind == 0:      # character-like names: two or three syllable words
            add("".join(_SYL[(b >> (5 * i)) % len(_SYL)] for i in range(2 + b % 2)) + "_" + "".join(_SYL[(c >> (5 * i)) % len(_SYL)] for i in range(2 + c % 3)))
        elif kind == 1:
            g = groups[b % len(groups)]
            add(_COLORS[c % len(_COLORS)] + "_" + g[d % len(g)] + "_" + str(b % 97))
        elif kind == 2:
            g = groups[b % len(groups)]
            add(g[c % len(g)] + "_(" + _MISC[d % len(_MISC)] + ")")
        elif kind == 3:
            add(_MISC[b % len(_MISC)] + "_" + _MISC[c % len(_MISC)] + "_" + str(d % 997))
        else:
            g = groups[b % len(groups)]
            add(g[c % len(g)] + "-" + _SYL[d % len(_SYL)] + str(b % 50))
    return out[:n]


def make_lexicon(vocab, n_entries, seed=2, chains=False):
    """Half aliases (keys outside the vocabulary), half implied lists.  With chains=True the aliases form chains of
    depth 2-8 plus some 2-cycles, so a tag advances one hop per NormalizeTag site (Program.cs:1155-1156)."""
    r = splitmix64(seed, 6 * n_entries + 16, stream=3)
    V = len(vocab)
    aliases, implied = {}, {}
    n_alias = n_entries // 2
    i = 0
    k = 0
    while len(aliases) < n_alias:
        if chains and (int(r[k]) % 4 == 0) and len(aliases) + 8 < n_alias:
            depth = 2 + int(r[k + 1]) % 7
            names = ["syn%d_%d" % (i, j) for j in range(depth)]
            for j in range(depth - 1):
                aliases[names[j]] = names[j + 1]
            aliases[names[-1]] = vocab[int(r[k + 2]) % V]
        elif chains and (int(r[k]) % 64 == 1):
            a, b = "cyc%d_a" % i, "cyc%d_b" % i
            aliases[a] = b
            aliases[b] = a
        else:
            aliases["aka%d" % i] = vocab[int(r[k + 2]) % V]
        i += 1
        k += 3
    # implied: keys from the vocabulary's tail, values from a small head of general tags that imply nothing further
    # (the shape of the shipped lexicon: barefoot -> feet, licking -> oral, mouth)
    head = min(256, V // 4)
    k = 3 * n_entries
    while len(implied) < min(n_entries - n_alias, V - head):
        key = vocab[head + int(r[k]) % (V - head)]
        m = 1 + int(r[k + 1]) % 2
        implied[key] = [vocab[(int(r[k + 2]) + 131 * j) % head] for j in range(m)]
        k += 3
        if k + 3 >= len(r):
            break
    return {"aliases": aliases, "implied": implied}


def make_dense_rules(vocab, n_rules=64, n_subs=4, n_facets=32, n_facet_rules=4, seed=3):
    r = splitmix64(seed, 16 * (n_rules * (n_subs + 2) + n_facets * (n_facet_rules + 1)) + 64, stream=5)
    V = min(len(vocab), 2000)          # literals from the frequent head so that rules do fire
    k = 0

    def pick(cnt):
        nonlocal k
        v = [vocab[int(r[k + j]) % V] for j in range(cnt)]
        k += cnt
        return v

    actions = []
    for i in range(n_rules):
        actions.append({
            "parent_tag": "act%d" % i,
            "required_all": pick(1 + i % 2),
            "required_any": pick(i % 3),
            "forbidden_any": pick(i % 2),
            "subtypes": [{"tag": "act%d_sub%d" % (i, j), "match_any": pick(2), "match_all": pick(j % 2)} for j in range(n_subs)],
        })
    facets = {}
    for i in range(n_facets):
        facets["facet%d" % i] = {"default": "unknown_facet%d" % i,
                                 "rules": [{"tag": "facet%d_v%d" % (i, j), "match_any": pick(2), "match_all": []} for j in range(n_facet_rules)]}
    return actions, facets


class Corpus:
    """arena/offsets plus the keyword arguments for imagecaptioner_amd.Config."""

    def __init__(self, name, arena, offsets, config_kwargs, pool):
        self.name = name
        self.arena = arena
        self.offsets = offsets
        self.config_kwargs = config_kwargs
        self.pool = pool            # the strings tags were drawn from

    @property
    def n_files(self):
        return len(self.offsets) - 1

    def line(self, i):
        return bytes(self.arena[int(self.offsets[i]):int(self.offsets[i + 1])])


def _pack_pool(pool):
    enc = [p.encode("ascii") for p in pool]
    lens = np.array([len(b) for b in enc], dtype=np.int64)
    starts = np.zeros(len(enc) + 1, dtype=np.int64)
    np.cumsum(lens, out=starts[1:])
    return np.frombuffer(b"".join(enc), dtype=np.uint8), starts, lens


def build_lines(pool, counts, picks, chunk_files=32768):
    """Lines "t, t, ..., t\\n" from per-file tag counts (all >= 1) and a flat array of pool indices (vectorised ragged gather)."""
    pbytes, pstart, plen = _pack_pool(pool)
    counts = np.asarray(counts, dtype=np.int64)
    assert counts.min() >= 1
    n = len(counts)
    tag_begin = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(counts, out=tag_begin[1:])
    arenas = []
    file_len = np.zeros(n, dtype=np.int64)
    for f0 in range(0, n, chunk_files):
        f1 = min(n, f0 + chunk_files)
        t0, t1 = int(tag_begin[f0]), int(tag_begin[f1])
        ids = picks[t0:t1]
        tl = plen[ids]
        is_last = np.zeros(t1 - t0, dtype=bool)
        is_last[tag_begin[f0 + 1:f1 + 1] - 1 - t0] = True
        sl = tl + np.where(is_last, 1, 2)                       # tag + ", "  or  tag + "\n"
        dst = np.zeros(len(sl) + 1, dtype=np.int64)
        np.cumsum(sl, out=dst[1:])
        total = int(dst[-1])
        rep = np.repeat(np.arange(len(sl), dtype=np.int64), sl)  # segment of every output byte
        within = np.arange(total, dtype=np.int64) - dst[rep]
        tlr = tl[rep]
        out = np.full(total, ord(" "), dtype=np.uint8)
        body = within < tlr
        out[body] = pbytes[(pstart[ids][rep] + within)[body]]
        at_sep = within == tlr
        out[at_sep] = np.where(is_last[rep][at_sep], ord("\n"), ord(","))
        arenas.append(out)
        file_len[f0:f1] = np.add.reduceat(sl, tag_begin[f0:f1] - t0)
    arena = np.concatenate(arenas) if arenas else np.zeros(0, np.uint8)
    offsets = np.zeros(n + 1, dtype=np.uint64)
    np.cumsum(file_len, out=offsets[1:])
    return arena, offsets


def _tag_counts(kind, n, seed):
    u = _uniform01(splitmix64(seed, n, stream=11))
    if kind == "poisson30" or kind == "poisson40":
        lam = 30.0 if kind == "poisson30" else 40.0
        # normal approximation of Poisson, rounded and clipped: deterministic in the seed
        v = _uniform01(splitmix64(seed, n, stream=12))
        z = np.sqrt(-2.0 * np.log(np.maximum(u, 1e-12))) * np.cos(2 * np.pi * v)
        return np.clip(np.rint(lam + np.sqrt(lam) * z), 1, 4 * lam).astype(np.int64)
    if kind == "longtail":
        # Pareto-like tail clipped to 500 tags
        return np.clip(np.floor(12.0 / np.power(1.0 - u * 0.99995, 0.75)), 1, 500).astype(np.int64)
    raise ValueError(kind)


def _picks(total, pool_n, seed, zipf_s=None, start=0):
    u = _uniform01(splitmix64(seed, total, stream=13, start=start))
    if zipf_s is None:
        return np.minimum((u * pool_n).astype(np.int64), pool_n - 1)
    w = 1.0 / np.power(np.arange(1, pool_n + 1, dtype=np.float64), zipf_s)
    cdf = np.cumsum(w)
    cdf /= cdf[-1]
    return np.minimum(np.searchsorted(cdf, u), pool_n - 1).astype(np.int64)


def make_corpus(name, n_files, seed=1234, file_range=None):
    """One of the benchmark configurations (c1..c5; c3c: c3 with the alias chains inside the vocabulary; c2n: c2 without a
    vocabulary file) as a Corpus.  file_range = (begin, end): only those files of the n_files-file corpus (byte for byte the
    slice of the whole corpus; the tag draws before `begin` are skipped, not made), which is how one rank of a sharded run gets
    its part of ONE corpus without building the rest."""
    name = name.lower()
    lo, hi = (0, n_files) if file_range is None else (int(file_range[0]), int(file_range[1]))
    if not (0 <= lo <= hi <= n_files):
        raise ValueError("file_range outside the corpus")

    def draw(kind, pool_n, zipf_s=None):
        counts = _tag_counts(kind, n_files, seed)              # every file's tag count (cheap), so that the slice knows where its tags begin
        first_tag = int(counts[:lo].sum())
        counts = counts[lo:hi]
        return counts, _picks(int(counts.sum()), pool_n, seed, zipf_s=zipf_s, start=first_tag)
    if name == "c1":
        with open(os.path.join(REFERENCE_DATA, "top_tags.txt"), "r", encoding="utf-8") as f:
            vocab = [t.strip() for t in f if t.strip()]
        pool = vocab
        counts, picks = draw("poisson30", len(pool), zipf_s=1.0)
        cfg = dict(lexicon=os.path.join(REFERENCE_DATA, "rules", "lexicon.json"), actions=os.path.join(REFERENCE_DATA, "rules", "actions.json"),
                   required_facets=os.path.join(REFERENCE_DATA, "rules", "required_facets.json"),
                   profile=os.path.join(REFERENCE_DATA, "profiles", "sgrui.json"), vocabulary=os.path.join(REFERENCE_DATA, "top_tags.txt"),
                   minimal_subject_tags="1girl, solo")
    elif name in ("c2", "c4", "c2n"):
        vocab = make_vocabulary(20000, seed=1)
        lex = make_lexicon(vocab, 20000, seed=2)
        pool = vocab + list(lex["aliases"].keys())[:2000]          # alias keys do occur in the input
        counts, picks = draw("poisson40", len(pool), zipf_s=1.1 if name == "c4" else None)
        cfg = dict(lexicon=json.dumps(lex).encode(), vocabulary=("\n".join(vocab) + "\n").encode(), vocabulary_is_json=0,
                   minimal_subject_tags="1girl, solo")
        if name == "c2n":
            # c2 without a vocabulary file: no strict vocabulary, so the tags in no table (a third of the pool) are kept
            cfg["vocabulary"] = None
    elif name in ("c3", "c3c"):
        vocab = make_vocabulary(50000, seed=1)
        lex = make_lexicon(vocab, 100000, seed=2, chains=True)
        pool = vocab + list(lex["aliases"].keys())[:20000]
        counts, picks = draw("poisson40", len(pool))
        # c3: the vocabulary file holds the plain tags only, so strict vocabulary (which a vocabulary file switches on,
        # Program.cs:999-1002) drops every chain member at ingest (Program.cs:601).  c3c: the chain and cycle members are
        # in the vocabulary file too, so a tag moves one alias hop at each NormalizeTag site and the chains are walked.
        known = vocab + ([k for k in lex["aliases"] if k.startswith(("syn", "cyc"))] if name == "c3c" else [])
        cfg = dict(lexicon=json.dumps(lex).encode(), vocabulary=("\n".join(known) + "\n").encode(), vocabulary_is_json=0,
                   minimal_subject_tags="1girl, solo")
    elif name == "c5":
        vocab = make_vocabulary(20000, seed=1)
        lex = make_lexicon(vocab, 20000, seed=2)
        actions, facets = make_dense_rules(vocab)
        pool = vocab + list(lex["aliases"].keys())[:2000]
        counts, picks = draw("longtail", len(pool), zipf_s=0.9)
        cfg = dict(lexicon=json.dumps(lex).encode(), actions=json.dumps(actions).encode(), required_facets=json.dumps(facets).encode(),
                   vocabulary=("\n".join(vocab) + "\n").encode(), vocabulary_is_json=0, minimal_subject_tags="1girl, solo")
    else:
        raise ValueError("unknown corpus " + name)
    if hi == lo:
        return Corpus(name, np.zeros(0, np.uint8), np.zeros(1, np.uint64), cfg, pool)
    arena, offsets = build_lines(pool, counts, picks)
    return Corpus(name, arena, offsets, cfg, pool)

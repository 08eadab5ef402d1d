/*
 * tagpipe.h — C ABI of the MI355X batch tag pipeline (libtagpipe_hip.so).
 *
 * The reference (Mumicanso8323/ImageCaptioner, Program.cs) has no FFI seam:
 * every function on the path is `private static` in one class.  This header
 * defines the seam a maintainer would P/Invoke: it wraps exactly the pure
 * span of the per-image loop body — Program.cs:223-229 and 292-329 with the
 * OpenAI branch off, fed by ParseTagCandidatesFromLine (Program.cs:574-605) —
 * and is called once per BATCH of files instead of once per image.
 *
 * Plain C types only; no callbacks; no exceptions cross the boundary.
 * Every function returns 0 on success or a negative TP_E_* code; the message
 * for the calling thread's last error is available from tp_last_error().
 * One context may be used by one thread at a time.
 */
#ifndef TAGPIPE_H_
#define TAGPIPE_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TP_OK              0
#define TP_E_INVALID_ARG  -1
#define TP_E_CONFIG       -2  /* malformed JSON / unsupported table content (reference: "Fatal error occurred.", Program.cs:371-375) */
#define TP_E_DEVICE       -3  /* HIP runtime failure, no device, kernel fault */
#define TP_E_CAPACITY     -4  /* caller-provided output buffer too small; required sizes are reported */
#define TP_E_NOMEM        -5

/* tp_verdict.coverage bits: 1 = category present (Program.cs:1058-1064, 1087-1092). */
#define TP_COV_IDENTITY    0x01u
#define TP_COV_HAIR        0x02u
#define TP_COV_EYES        0x04u
#define TP_COV_OUTFIT      0x08u
#define TP_COV_POSE_ACTION 0x10u
#define TP_COV_BACKGROUND  0x20u

/* tp_verdict.flags bits. */
#define TP_F_LOW           0x01u  /* MissingCoreBuckets(finalTags), Program.cs:319, 1097-1101 */
#define TP_F_ELIGIBLE      0x02u  /* evidence has feet and penis, Program.cs:309 */
#define TP_F_FOOTJOB       0x04u  /* ... and the final tags have footjob, Program.cs:311 */
#define TP_F_FALLBACK      0x08u  /* fewer than 3 selected tags: minimal fallback line used, Program.cs:297-298 */
#define TP_F_LINE_REJECTED 0x10u  /* NormalizeRawTagLine returned "" (blank or LooksBad), Program.cs:590, 596 */
#define TP_F_UNSUPPORTED   0x20u  /* input this build does not process; nothing is written for the file.  Text outside ASCII is processed (Trim,
                                     \s and ToLowerInvariant over UTF-16 as in Program.cs:1150-1154, characters beyond the Basic Multilingual Plane
                                     copied as they are); what is refused: text that is not well-formed UTF-8 (the reference would read U+FFFD
                                     there); a lower-case character that shares its capital with another one (medial and final sigma, micro sign
                                     and mu, ...) in a tagger reply, or in a line when some table string holds such a character, because the
                                     reference's ordinal-ignore-case sets take two such spellings for one tag; a line that lower-casing makes
                                     longer than its buffer (U+023A, U+023E); a per-file list that outgrew its workspace */
#define TP_F_SAFE_REJECT   0x40u  /* DetectSafeModeRejectReason hit, Program.cs:329, 942-945 */
#define TP_F_TRIAGE_BAD    0x80u  /* tp_triage_lines only: LooksBad(NormalizeRawTagLine(text)), Program.cs:206-207 */

/* Candidate sources and their SourceRank (Program.cs:398, 712-717). */
#define TP_SRC_PRIMARY_JOY   0
#define TP_SRC_SECONDARY_JOY 1
#define TP_SRC_OPENAI        2

/*
 * Per-file verdict record, 16 bytes.  The reference keeps these facts in
 * locals of the loop body (coverage: Program.cs:227; isLow: :319; counters:
 * :309-312; rejectReason: :329) and in the lists it hands to AppendAudit
 * (:334); this record is their fixed-size part.
 */
typedef struct tp_verdict {
    uint8_t  coverage;      /* TP_COV_* of EvaluateCoverage(finalSelection.Tags) */
    uint8_t  flags;         /* TP_F_* */
    uint16_t n_tags;        /* tags in the emitted caption line */
    uint16_t n_candidates;  /* finalCandidates.Count */
    uint16_t n_selected;    /* finalSelection.Tags.Count (before the fallback) */
    int16_t  safe_hit;      /* index into the SafeModeRejectTags list (Program.cs:43-45 order) of the hit, else -1 */
    uint16_t n_evidence;    /* evidenceTags.Count after ExpandImpliedTags; 0xFFFF (not counted) when the call did not ask for the
                               detail arrays, i.e. tp_device_io.fired == NULL (tp_run_lines / tp_run_scored always count) */
    uint16_t n_unknown;     /* GetUnknownTags(finalTags).Count, Program.cs:314 */
    uint16_t n_dropped;     /* finalSelection.DroppedTagsWithReason.Count */
} tp_verdict;

/* Batch-wide counters (the Interlocked counters of Program.cs:194-195 plus build-side status counts). */
typedef struct tp_counters {
    uint64_t files;
    uint64_t low;
    uint64_t action_eligible;
    uint64_t action_footjob;
    uint64_t fallback;
    uint64_t line_rejected;
    uint64_t unsupported;
    uint64_t safe_rejected;
} tp_counters;

/*
 * Configuration, mirroring what Main loads before the loop (Program.cs:156-165).
 * Each table source is given either as in-memory bytes (data/len, wins when
 * data != NULL) or as a path (NULL or a missing file = table absent, as the
 * reference's File.Exists checks).  Formats are the reference's own.
 */
typedef struct tp_source {
    const char*    path;
    const uint8_t* data;
    uint64_t       len;
} tp_source;

typedef struct tp_config {
    uint32_t  struct_size;            /* sizeof(tp_config), for ABI growth */
    tp_source lexicon;                /* rules/lexicon.json,         LoadLexicon        Program.cs:1275-1305 */
    tp_source actions;                /* rules/actions.json,         LoadActions        Program.cs:1307-1338 */
    tp_source required_facets;        /* rules/required_facets.json, LoadRequiredFacets Program.cs:1374-1406 */
    tp_source profile;                /* profiles/<name>.json,       LoadProfile        Program.cs:918-931  */
    tp_source vocabulary;             /* top_tags.txt or .json,      LoadKnownTagDictionary Program.cs:983-1053 */
    int32_t   vocabulary_is_json;     /* 1 = parse as JSON (the reference decides by the ".json" extension, Program.cs:1016-1017);
                                         0 = text; -1 = decide from vocabulary.path's extension */
    tp_source allowlist;              /* allowlist.txt, LoadTagFile Program.cs:1429-1439 */
    tp_source denylist;               /* denylist.txt */
    const char* trigger;              /* --trigger, default "sgrui" (Program.cs:1540) */
    const char* auto_prefix_tags;     /* --auto-prefix-tags   (Program.cs:158, 165) */
    const char* minimal_subject_tags; /* --minimal-subject-tags (Program.cs:165) */
    const char* newline;              /* Environment.NewLine written after each caption (Program.cs:317): "\n" or "\r\n"; NULL = "\n" */
    int32_t   no_profile;             /* --no-profile */
    int32_t   strict_known;           /* --strict-known-tags value when explicit */
    int32_t   strict_known_explicit;  /* either strict flag was given (Program.cs:999-1002, 1538) */
    int32_t   safe_mode;              /* default on (Program.cs:1543) */
    double    line_weight;            /* sourceWeight for line ingest; the self-check uses 1.0 (Program.cs:1958) */
    int32_t   line_source;            /* TP_SRC_*; default parameter is OpenAI (Program.cs:574) */
    int32_t   device;                 /* HIP device ordinal */
} tp_config;

/* Fill *cfg with the reference's defaults (safe mode on, trigger "sgrui", weight 1.0, source OpenAI, "\n"). */
void tp_config_init(tp_config* cfg);

typedef struct tp_ctx tp_ctx;

/*
 * Load and compile the tables in the reference's order (Program.cs:158-165:
 * allow/deny lists, vocabulary, profile, force-allow set, then the rule
 * engine JSONs) and upload them to the device.  Returns NULL on failure.
 */
tp_ctx* tp_create(const tp_config* cfg);
void    tp_destroy(tp_ctx* ctx);
const char* tp_last_error(void);

/* Facts about the compiled tables (sizes of the optional detail arrays below). */
typedef struct tp_table_info {
    uint32_t n_strings;        /* interned strings (histogram length) */
    uint32_t n_action_rules;   /* ActiveActionRules.Length */
    uint32_t n_action_subtypes;/* total subtypes over all rules */
    uint32_t n_facets;         /* ActiveRequiredFacets.Length */
    uint32_t fired_words;      /* u32 words per file in the fired-rule bitmap: ceil((rules + subtypes) / 32) */
    uint32_t n_known_tags;     /* KnownTags.Count */
    uint32_t strict_known;     /* ActiveStrictKnownTags after load */
    uint32_t known_found;      /* KnownTagDictionaryFound */
} tp_table_info;
int tp_get_table_info(const tp_ctx* ctx, tp_table_info* out);

/* The interned string with the given id (histogram index -> tag).  Pointer valid until tp_destroy. */
int tp_get_string(const tp_ctx* ctx, uint32_t id, const char** utf8, uint32_t* len);

/*
 * Names behind the detail arrays of tp_result, i.e. the strings the reference
 * logs as "action:{parent}", "subtype:{parent}:{tag}", "facet:{name}:{tag}"
 * and "facet:{name}:{default}:default" (Program.cs:741, 748, 764, 772).
 */
#define TP_NAME_ACTION_RULE    0  /* index = rule           -> ParentTag */
#define TP_NAME_ACTION_SUBTYPE 1  /* index = global subtype -> "parent:tag" */
#define TP_NAME_FACET          2  /* index = facet          -> Name */
#define TP_NAME_FACET_DEFAULT  3  /* index = facet          -> DefaultTag */
#define TP_NAME_FACET_RULE     4  /* index = facet << 16 | rule -> rule Tag */
#define TP_NAME_SAFE_TAG       5  /* index = tp_verdict.safe_hit -> the SafeModeRejectTags entry (Program.cs:43-45) */
int tp_get_name(const tp_ctx* ctx, int kind, uint32_t index, const char** utf8, uint32_t* len);

/*
 * Result of one batch, owned by the library until tp_release.
 * captions[caption_offsets[i] .. caption_offsets[i+1]) is file i's caption line
 * including the configured newline: the bytes File.WriteAllTextAsync writes at
 * Program.cs:317.
 */
typedef struct tp_result {
    uint64_t        n_files;
    const uint8_t*  captions;
    const uint64_t* caption_offsets;   /* n_files + 1 */
    const tp_verdict* verdicts;        /* n_files */
    tp_counters     counters;
    const uint64_t* histogram;         /* n_strings counts: files whose caption holds the interned tag (build-defined, no reference counterpart) */
    uint64_t        histogram_other;   /* caption tags that are not interned strings */
    /* Detail arrays for the audit consumers (Program.cs:334): */
    const uint32_t* fired;             /* n_files * fired_words: bit r = action rule r fired; bit n_action_rules + s = subtype s (global index) fired (Program.cs:741, 748) */
    const uint16_t* facet_choice;      /* n_files * n_facets: rule index chosen, 0xFFFF = default tag (Program.cs:760-773) */
    void*           opaque;
} tp_result;

/*
 * Line ingest (ParseTagCandidatesFromLine, Program.cs:574-587): file i's raw
 * text is arena[offsets[i] .. offsets[i+1]).  Host pointers; the call uploads,
 * runs the kernels and downloads the result.
 *
 * Tags in no table.  When no strict vocabulary drops them (no vocabulary file,
 * or --strict-known-tags off: Program.cs:999-1002, 1131) such tags are kept,
 * and a file that holds one takes the slow thread-per-file kernel.  This call
 * and tp_run_lines_device therefore note the plain strings of that kind they
 * meet ([a-z0-9_()!/-], up to 32 bytes) and, when the run is over, compile the
 * context's tables again with those strings added, so that later calls find
 * them.  Nothing a caller can see changes by that: results are the same
 * before and after, the configuration's strings keep their ids, n_strings
 * and the histogram's length stay as tp_get_table_info reported them, and a
 * learned tag is counted in histogram_other like any tag in no table.  The
 * compile takes a few tenths of a second and happens only after a call that
 * met new strings.  TAGPIPE_NO_LEARN=1 in the environment turns this off.
 */
int tp_run_lines(tp_ctx* ctx, const uint8_t* arena, const uint64_t* offsets, uint64_t n_files, tp_result* out);

/*
 * Scored ingest (candidates shaped like ParseJoyTags output, Program.cs:463-489):
 * file i owns tags tag_begin[i] .. tag_begin[i+1]; tag t's raw text is
 * arena[tag_offsets[t] .. tag_offsets[t+1]), with score scores[t] and source
 * sources[t] (TP_SRC_*).  Each tag is normalised once, blank ones are skipped,
 * order is the running index, and the list is sorted by score descending then
 * order (Program.cs:483-487) before SelectTags.  Scores must be finite: a NaN
 * or an infinity anywhere in `scores` fails the call with TP_E_INVALID_ARG
 * before anything runs (the reference's JSON ingest cannot produce either, and
 * its comparisons and Math.Max / Math.Abs steps are left undefined for them
 * here); tp_run_merged does the same.  Everything finite is taken as it is:
 * negative, above 1, -0.0, denormal.
 */
int tp_run_scored(tp_ctx* ctx, const uint8_t* arena, const uint64_t* tag_offsets, const uint64_t* tag_begin,
                  const double* scores, const uint8_t* sources, uint64_t n_files, tp_result* out);

/*
 * Scored ingest of the two tagger replies the reference asks for per image
 * (Program.cs:214-221: primary threshold, then --joy-threshold-secondary):
 * file i owns tags tag_begin[i] .. tag_begin[i+1], of which those below
 * secondary_begin[i] are the primary reply (TP_SRC_PRIMARY_JOY) and the rest
 * the secondary one (TP_SRC_SECONDARY_JOY); tag_begin[i] <= secondary_begin[i]
 * <= tag_begin[i+1].  Each reply is parsed like ParseJoyTags (Program.cs:463-489),
 * the secondary one is cut down by LimitSecondaryTags with secondary_cap new
 * tags (Program.cs:801-815; the reference's default is 15, <= 0 drops the
 * whole reply), MergeCandidateTags (Program.cs:442-461) joins them, and the
 * merged list goes through the pipeline.
 */
int tp_run_merged(tp_ctx* ctx, const uint8_t* arena, const uint64_t* tag_offsets, const uint64_t* tag_begin,
                  const uint64_t* secondary_begin, const double* scores, int secondary_cap, uint64_t n_files, tp_result* out);

void tp_release(tp_ctx* ctx, tp_result* res);

/*
 * Cached tagger replies into the arrays tp_run_scored / tp_run_merged take: the JSON bodies JoyTag's /tag endpoint
 * returned for every file (CallJoyTagAsync's "tags" lookup and ParseJoyTags, Program.cs:463-489, 500-505).  File i's
 * primary reply is primary[primary_offsets[i] .. primary_offsets[i + 1]) (UTF-8 JSON); secondary / secondary_offsets hold
 * the second call's replies (Program.cs:219) or are NULL.  Out: the tags' texts as the replies spell them (the run normalises
 * them), tag_offsets, tag_begin and - with secondary replies - secondary_begin in tp_run_merged's layout, scores (a string
 * element's score is max(0.01, 1.0 - order * 0.001), an object's its "score", else "confidence", else 0.5; "tag" else "name"),
 * sources (TP_SRC_PRIMARY_JOY / TP_SRC_SECONDARY_JOY by reply).  Tags that NormalizeTag makes blank are left out, as
 * ParseJoyTags leaves them out (they take no Order).  status[i] is the worse of the file's replies:
 */
#define TP_REPLY_OK       0
#define TP_REPLY_NO_TAGS  1  /* well-formed JSON without a "tags" array: an empty list, as in the reference (Program.cs:502-503) */
#define TP_REPLY_ORDER    2  /* no element carries a score, and the default scores rise somewhere along the list (string and score-less
                                object elements mixed): ParseJoyTags keeps such a list unsorted (Program.cs:487).  The tags are returned
                                in the reply's order; tp_run_replies keeps that order (has_score says which replies were sorted),
                                tp_run_scored / tp_run_merged sort every list, which gives the same captions, verdicts, counters and
                                fired rules (SelectTags sorts by score itself, Program.cs:651-655) but lists the evidence and the
                                dropped tags of the audit row in sorted instead of reply order */
#define TP_REPLY_FAILED   3  /* the reference throws on it (JSON that does not parse, a root that is not an object, a "score" or
                                "confidence" that is not a number): the image fails there; returned empty */
typedef struct tp_replies {
    uint64_t n_files, n_tags;
    const uint8_t*  arena;
    const uint64_t* tag_offsets;      /* n_tags + 1 */
    const uint64_t* tag_begin;        /* n_files + 1 */
    const uint64_t* secondary_begin;  /* n_files, NULL without secondary replies */
    const double*   scores;           /* n_tags */
    const uint8_t*  sources;          /* n_tags */
    const uint8_t*  status;           /* n_files, TP_REPLY_* */
    const uint8_t*  has_score;        /* n_files: bit 0 the primary reply's HasScore, bit 1 the secondary's (Program.cs:463) */
    void* opaque;
} tp_replies;
int tp_decode_joytag_replies(tp_ctx* ctx, const uint8_t* primary, const uint64_t* primary_offsets, const uint8_t* secondary,
                             const uint64_t* secondary_offsets, uint64_t n_files, tp_replies* out);
void tp_release_replies(tp_replies* replies);
/*
 * The decoded replies through the pipeline: what Program.cs:214-224 does with CallJoyTagAsync's lists.  Without secondary
 * replies (replies->secondary_begin NULL) every file's list goes to SelectTags as ParseJoyTags left it: sorted by (score
 * descending, order) when an element carried a score, in the reply's own order otherwise (Program.cs:487; has_score bit 0).
 * With secondary replies the two lists are merged as tp_run_merged does (secondary_cap = --secondary-new-tags-cap), the
 * primary one again in ParseJoyTags' order.  Files whose status is TP_REPLY_FAILED hold no tags: the reference fails such
 * an image before the pipeline, and the caller does the same with what comes back for them.  `replies` may be the struct
 * tp_decode_joytag_replies filled or one the caller built with the same layout (host pointers).
 */
int tp_run_replies(tp_ctx* ctx, const tp_replies* replies, int secondary_cap, tp_result* out);

/*
 * Device-resident variant of tp_run_lines for callers that already hold the
 * corpus in HBM (and for benchmarking the kernels without PCIe): every pointer
 * is a device pointer on the context's device; the caller provides the output
 * buffers.  histogram (n_strings u64) and counters are ACCUMULATED into, so a
 * caller can shard one corpus over several calls (or all-reduce them across
 * GPUs).  fired / facet_choice may be NULL.  Work is enqueued on `stream`
 * (a hipStream_t, NULL = default stream); the call synchronises the stream
 * once to learn the caption size.  If caption_capacity is too small the call
 * returns TP_E_CAPACITY and *caption_bytes holds the required size.  Files are
 * worked through in ranges of up to 2^24: by the time a later range finds the
 * caption buffer full, earlier ranges have written their captions and added
 * their tags to histogram and counters.  After TP_E_CAPACITY the outputs are
 * therefore unspecified, and a caller that retries with a larger buffer must
 * set histogram and counters back to what they held before the failed call.
 */
typedef struct tp_device_io {
    const uint8_t*  arena;            /* in; 16 bytes past offsets[n_files] must be readable (the kernels fetch texts in aligned pieces; zeros there keep the last file on the fastest path) */
    const uint64_t* offsets;          /* in, n_files + 1 */
    uint64_t        n_files;
    uint8_t*        captions;         /* out */
    uint64_t        caption_capacity;
    uint64_t*       caption_offsets;  /* out, n_files + 1 */
    tp_verdict*     verdicts;         /* out, n_files */
    uint64_t*       histogram;        /* in/out, n_strings + 1 (last = other), may be NULL */
    tp_counters*    counters;         /* in/out, may be NULL */
    uint32_t*       fired;            /* out, may be NULL */
    uint16_t*       facet_choice;     /* out, may be NULL */
    void*           stream;
} tp_device_io;
int tp_run_lines_device(tp_ctx* ctx, const tp_device_io* io, uint64_t* caption_bytes);

/*
 * tp_run_lines_device for callers that cannot allocate device memory while they
 * run (tight serving loops, harnesses that forbid hipMalloc).  No counterpart in
 * the reference.  Same arguments, outputs and return codes, but the call makes no
 * hipMalloc / hipFree / event creation (it does synchronise the stream, so it
 * cannot itself be recorded into a HIP graph):
 * everything it needs besides the caller's buffers lives in one static array
 * of 16 MiB that each GPU the library runs on holds until the process ends.
 * Limits: at most TP_NOALLOC_MAX_FILES files a call (more: TP_E_INVALID_ARG);
 * files that need the general kernel share about 6 MiB of workspace, 64 files
 * at a time at the least (does not fit: TP_E_NOMEM, nothing useful written).
 * tp_get_last_timing reports zeros after this call.  Every context on a GPU
 * shares that GPU's array: calls that overlap in time on one GPU race, also
 * from different contexts or streams; the array keeps the last call's contents
 * and the call clears what it needs cleared.  The call does not compile tables
 * again (see tp_run_lines on tags in no table): what it meets is noted as a
 * copy of the tag's bytes (never as a pointer into the caller's arena, which may
 * be freed or reused as soon as the call returns) and learned by the next
 * tp_run_lines / tp_run_lines_device call on the context.  A context whose GPU's
 * array cannot be bound (device ordinal of 64 or more) reports TP_E_DEVICE here
 * and works normally through every other entry point.
 */
#define TP_NOALLOC_MAX_FILES 16384
int tp_run_lines_device_noalloc(tp_ctx* ctx, const tp_device_io* io, uint64_t* caption_bytes);

/*
 * Existing-caption triage (Program.cs:203-210 with --reprocess-bad-existing): for every file's text, would the
 * reference regenerate the caption?  bad[i] = 1 when LooksBad(NormalizeRawTagLine(text_i)) (the file is deleted and the
 * image captioned again, Program.cs:207-209), 0 when the existing caption is kept (the image is skipped), 2 when the text
 * is something this build does not process (see TP_F_UNSUPPORTED).  Host pointers; bad has n_files entries.
 */
int tp_triage_lines(tp_ctx* ctx, const uint8_t* arena, const uint64_t* offsets, uint64_t n_files, uint8_t* bad);

/*
 * Corpus-wide tag table in the format the reference reads as top_tags.txt
 * (one tag per line, Program.cs:1021-1028), ordered by count descending then
 * by the tag's bytes ascending; tags with count 0 are omitted.  Build-defined:
 * the reference never writes this file.  `histogram` is host memory of
 * n_strings counts.  Writes at most `capacity` bytes to `out`; *needed is the
 * full size.
 */
int tp_format_top_tags(const tp_ctx* ctx, const uint64_t* histogram, uint8_t* out, uint64_t capacity, uint64_t* needed);

/*
 * The caption the reference writes for an image whose processing threw
 * (Program.cs:344-350): string.Join(", ", SelectTags(ParseTagCandidatesFromLine(
 * minimalFallback, null, 1.0, TagSource.OpenAI)).Tags) + the configured newline;
 * the tags are joined as SelectTags returns them (no FormatCaptionSingleLine pass).
 * The same text for every image of a run: the caller writes it for a file whose
 * verdict carries TP_F_UNSUPPORTED (text this library does not process) or whose
 * own stages before the call failed, when <image>.txt does not exist yet, and
 * "error; check later" + newline to <image>.needtag under --keep-needtag
 * (tp_error_needtag_text()).  Writes at most `capacity` bytes; *needed is the
 * full size (TP_E_CAPACITY when capacity is smaller; nothing is written then).
 */
int tp_fallback_caption(const tp_ctx* ctx, uint8_t* out, uint64_t capacity, uint64_t* needed);
const char* tp_error_needtag_text(void);

/* Timing of the most recent run on this context (milliseconds, from HIP events on the run's stream). */
typedef struct tp_timing {
    float total_ms;     /* first kernel start to last kernel end */
    float select_ms;    /* all wave-per-file kernels: the lean stage (k_lean_a, k_lean_b) and the stages behind it (k_fast, k_fast2, the long-file kernels) */
    float general_ms;   /* thread-per-file pipeline kernel (k_pipeline) over the files every wave-per-file stage declined */
    float emit_ms;      /* caption bytes, histogram, counters (k_emit) */
    float other_ms;     /* sizing, scans, host round trips between kernels */
    uint64_t general_files; /* how many files took the thread-per-file kernel */
    float lean_front_ms; /* part of select_ms: k_lean_a (scan of the line, table lookup of every tag, Distinct) */
    float lean_back_ms;  /* part of select_ms: k_lean_b (selection, implied tags, rules, facets, the final tag ids) */
    float fused_ms;      /* k_lean_bx, the plain run's fused kernel: k_lean_b's work, the caption offsets and the caption bytes of the
                            wave-per-file kernels' files (not part of select_ms or emit_ms; lean_back_ms is 0 then) */
    uint64_t fused_reruns; /* ranges (since tp_create) that k_lean_bx handed back to the separate kernels */
} tp_timing;
/*
 * firedRules of every file as AppendAudit joins them (Program.cs:334, 966:
 * string.Join(";", firedRules), before Csv()): "action:{parent}" and
 * "subtype:{parent}:{tag}" in rule order (Program.cs:735-749), then per facet
 * "facet:{name}:{tag}" for the chosen rule and "facet:{name}:{default}:default"
 * when the chosen tag is the default tag (Program.cs:760-772).  fired and
 * facet_choice are tp_result's detail arrays (host pointers).  File i's text is
 * out[out_offsets[i] .. out_offsets[i+1]).  *needed receives the total size;
 * TP_E_CAPACITY when capacity is smaller (nothing is written to out then).
 */
int tp_format_fired_rules(const tp_ctx* ctx, const uint32_t* fired, const uint16_t* facet_choice, uint64_t n_files,
                          uint8_t* out, uint64_t capacity, uint64_t* out_offsets, uint64_t* needed);

/*
 * What the loop writes beside a caption (Program.cs:319-332, 947-956), from
 * the verdicts (host pointer).  needtag[i]: TP_NEEDTAG_WRITE = write
 * tp_needtag_text() + the configured newline to <image>.needtag (low caption
 * and --keep-needtag), TP_NEEDTAG_DELETE = TryDelete it (not low),
 * TP_NEEDTAG_LEAVE = neither (low without --keep-needtag, or a file the device
 * code did not process).  reject[reject_offsets[i] .. reject_offsets[i+1]) is
 * the content of <image>.reject.txt, "safe_mode:{tag}" + newline, for files
 * with TP_F_SAFE_REJECT and empty for the others; the tag is spelled as in
 * SafeModeRejectTags (the reference prints the final tag's own spelling, which
 * differs only when a rule file writes such a tag with capitals).
 */
#define TP_NEEDTAG_DELETE 0
#define TP_NEEDTAG_WRITE  1
#define TP_NEEDTAG_LEAVE  2
const char* tp_needtag_text(void);
int tp_format_side_files(const tp_ctx* ctx, const tp_verdict* verdicts, uint64_t n_files, int keep_needtag, uint8_t* needtag,
                         uint8_t* reject, uint64_t capacity, uint64_t* reject_offsets, uint64_t* needed);

/*
 * The audit row (AppendAudit, Program.cs:334, 958-969).  With tp_set_audit(ctx, 1) the host-pointer runs (tp_run_lines,
 * tp_run_scored, tp_run_merged) also keep, per file, the three tag lists the row joins with ";" (Program.cs:966):
 *   TP_AUDIT_KEPT      keptTags  = finalTags as they stand after EnsureRequiredFacets (Program.cs:307), i.e. before
 *                                  FormatCaptionSingleLine normalises them again
 *   TP_AUDIT_EVIDENCE  evidenceTags after ExpandImpliedTags (Program.cs:301-302), in insertion order
 *   TP_AUDIT_DROPPED   finalSelection.DroppedTagsWithReason: "{tag}:invalid_tag" / "{tag}:policy_drop" (Program.cs:630-631, 682)
 * and two lists of the review record (BuildReviewRecord, Program.cs:826-868), entries separated by "\n":
 *   TP_AUDIT_TOP       TopJoyTags: the tags of the 20 best raw candidates by score, then order (Program.cs:860)
 *   TP_AUDIT_UNKNOWN   the caption's tags outside the vocabulary, in caption order (GetUnknownTags before its sort, Program.cs:817-821)
 * Column c of file i is text[offsets[TP_AUDIT_COLS * i + c] .. offsets[TP_AUDIT_COLS * i + c + 1]); tp_get_audit's pointers live
 * as long as the result.  In audit mode every wave-per-file stage keeps these lists for the files it takes, tag lines and
 * tagger replies alike, which costs about half a run's speed; of the replies the thread-per-file kernel is left with those
 * that kept their own order (TP_REPLY_ORDER), those with a tag in no table, and a single reply in which scores tie between
 * tags of different source ranks or between a tag it holds twice and another.  n_selected in the verdicts is exact in every mode;
 * n_evidence is counted.
 */
#define TP_AUDIT_KEPT     0
#define TP_AUDIT_EVIDENCE 1
#define TP_AUDIT_DROPPED  2
#define TP_AUDIT_TOP      3
#define TP_AUDIT_UNKNOWN  4
#define TP_AUDIT_COLS     5
typedef struct tp_audit {
    const uint8_t*  text;
    const uint64_t* offsets;   /* TP_AUDIT_COLS * n_files + 1 */
} tp_audit;
int tp_set_audit(tp_ctx* ctx, int on);
int tp_get_audit(const tp_result* res, tp_audit* out);

/* First line of __review/audit.csv, without a newline (Program.cs:964). */
const char* tp_audit_header(void);

/*
 * The lines AppendAudit appends to __review/audit.csv, one per file in file order (Program.cs:966-967): every field through
 * Csv() (Program.cs:976-981), joined with ",", followed by the configured newline.  names[name_offsets[i] ..
 * name_offsets[i + 1]) is Path.GetFileName(imagePath) of file i (UTF-8); vocab_file is KnownTagDictionaryPath (NULL when no
 * vocabulary file was found: the field is then the empty quoted string); used_profile is "none" under --no-profile, else the
 * profile's name (Program.cs:334).  coverage_stop_reason and coverage_trace are those of a run without the OpenAI passes
 * (Program.cs:226-228, 246-248, 292-294: one "pass0_joy ..." entry).  reject_reason spells the tag as SafeModeRejectTags
 * does (see tp_format_side_files).  A file with TP_F_UNSUPPORTED gets no line.  File i's line is out[out_offsets[i] ..
 * out_offsets[i + 1]); *needed receives the total size; TP_E_CAPACITY when capacity is smaller (nothing is written then).
 * res must come from a run in audit mode.
 */
int tp_format_audit_rows(const tp_ctx* ctx, const tp_result* res, const uint8_t* names, const uint64_t* name_offsets, const char* vocab_file,
                         const char* used_profile, uint8_t* out, uint64_t capacity, uint64_t* out_offsets, uint64_t* needed);

/*
 * The review log (BuildReviewRecord, AppendReviewLog, Program.cs:826-890): per file the line appended to
 * review_candidates.jsonl (format TP_REVIEW_JSONL: JsonSerializer.Serialize of the record, Program.cs:874, properties in the
 * order of Program.cs:415-431) or to review_candidates.csv (TP_REVIEW_CSV, Program.cs:875-881; tp_review_csv_header() is its
 * first line, Program.cs:886), each followed by the configured newline, and empty for files BuildReviewRecord returns null
 * for (Program.cs:845) and for files with TP_F_UNSUPPORTED.  image_paths / caption_paths are imagePath and txtPath of every
 * file (UTF-8, with their own offsets arrays); thresholds_used is the record's ThresholdsUsed text (Program.cs:847-849);
 * vocab_file is KnownTagDictionaryPath or NULL.  Which banned and subject tokens the raw candidates held comes from the
 * device (per-string facts compiled at load).  rawCandidates are the
 * candidates the run was given (the reference passes joyCandidates, which are the final candidates when the OpenAI passes
 * are off).  res must come from a run in audit mode.  Output conventions as tp_format_audit_rows.
 */
#define TP_REVIEW_JSONL 0
#define TP_REVIEW_CSV   1
const char* tp_review_csv_header(void);
int tp_format_review_rows(const tp_ctx* ctx, const tp_result* res, int format, const uint8_t* image_paths, const uint64_t* image_offsets,
                          const uint8_t* caption_paths, const uint64_t* caption_offsets, const char* thresholds_used, const char* vocab_file,
                          uint8_t* out, uint64_t capacity, uint64_t* out_offsets, uint64_t* needed);

int tp_get_last_timing(const tp_ctx* ctx, tp_timing* out);

#ifdef __cplusplus
}
#endif
#endif /* TAGPIPE_H_ */

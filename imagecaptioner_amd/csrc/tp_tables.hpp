// Host-side table compiler: reads the reference's configuration files in the
// reference's order (Program.cs:156-165) and compiles them into the flat table
// image the kernels read (tp_common.h: DevTables).
#pragma once
#include <cstdint>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/tagpipe.h"
#include "tp_common.h"

namespace tp {

struct ConfigError : std::runtime_error {
    explicit ConfigError(const std::string& m) : std::runtime_error(m) {}
};

struct HostTables {
    // arrays of the device image
    std::vector<uint8_t> str_bytes;
    std::vector<uint32_t> str_off, flags, next, alias1, lit, imp_off, imp_ids;
    std::vector<uint64_t> ht_hash;
    std::vector<uint32_t> ht_id;
    std::vector<uint16_t> dfa_next, dfa_out;
    std::vector<uint32_t> pool;
    std::vector<DRule> rules;
    std::vector<DSub> subs;
    std::vector<DFacet> facets;
    std::vector<DSub> facet_rules;
    std::vector<Rec> rec;
    std::vector<Slot> slots;
    std::vector<LeanSlot> lslots;           // empty when the two-choice placement failed
    std::vector<uint32_t> attr2, crec, fw_tix;
    std::vector<uint32_t> case_map;         // DevTables::case_map
    std::vector<uint32_t> emit_rec;         // DevTables::emit_rec
    std::vector<uint32_t> emit_rec32;       // DevTables::emit_rec32
    std::vector<uint32_t> lkey_hi;          // DevTables::lkey_hi
    std::vector<uint8_t> basic_fixed;       // per id: the string is a fixed point of the normalisation before the alias lookup
    std::vector<HtInfo> ht_info;
    std::vector<uint64_t> str_words;
    std::vector<uint32_t> step_off, step_ids, reach_off, reach_ent, rule_cache;
    DevTables scalars;                      // scalar members filled, pointer members null
    // host-side facts
    std::vector<std::string> strings;       // UTF-8 text per interned id
    std::vector<std::string> rule_names;    // ActionRule.ParentTag per rule
    std::vector<std::string> sub_names;     // "parent:tag" per subtype (global index)
    std::vector<std::string> facet_names, facet_defaults;
    std::vector<std::vector<std::string>> facet_rule_tags;
    std::vector<std::string> safe_tags;     // SafeModeRejectTags in the order tp_verdict.safe_hit indexes
    std::vector<uint8_t> fallback_line;     // the minimal fallback line (Program.cs:165), UTF-8
    uint32_t n_known = 0;
    bool strict = false, known_found = false;
};

// Throws ConfigError (malformed JSON or table content this build cannot represent).
// learned: strings met in the input that are in none of the configuration's tables (UTF-8; see tp_api.cpp), interned behind
// every string of the configuration (ids below DevTables::n_public are the same with and without them) with the attributes
// the pipeline computes for such a string; one that an interned string already covers is skipped.
void compile_tables(const tp_config& cfg, HostTables& out, const std::vector<std::string>* learned = nullptr);

// Stores the SelectTags result of the minimal fallback line (ids, evaluated by the pipeline kernel at load time) and
// derives the final-set form the wave-per-file kernel starts from.
void set_fallback(const HostTables& ht, DevTables& dt, const uint32_t* ids, uint32_t n);

// a < b under StringComparer.OrdinalIgnoreCase (UTF-16 code units, upper-cased the invariant way), both UTF-8.
bool less_ordinal_ignore_case(const std::string& a, const std::string& b);

// The corpus-wide tag table as a file LoadKnownTagDictionary reads (one tag per line, Program.cs:1021-1028): tags with a
// non-zero count, by count descending, then by the tag's bytes ascending.  `histogram` has one count per interned string.
std::string top_tags_text(const HostTables& ht, const uint64_t* histogram);

// Cached tagger replies (the JSON bodies of JoyTag's /tag endpoint) as the arrays tp_run_scored / tp_run_merged take:
// the "tags" lookup of CallJoyTagAsync and ParseJoyTags (Program.cs:463-489, 500-505).  One call appends one reply.
struct DecodedReplies {
    std::vector<uint8_t> arena;             // the tags' texts as the reply spells them, back to back
    std::vector<uint64_t> tag_offsets{0};   // into arena, one more than tags
    std::vector<double> scores;
};
// status (TP_REPLY_*), has_score: ParseJoyTags' HasScore.  A reply that fails or whose order cannot be kept appends nothing.
void decode_joytag_reply(const HostTables& ht, const std::unordered_map<std::string, uint32_t>& string_ids, const uint8_t* json, size_t len,
                         DecodedReplies& out, uint8_t& status, uint8_t& has_score);
std::unordered_map<std::string, uint32_t> index_strings(const HostTables& ht);

// A DevTables whose pointers address the HostTables vectors themselves (host-side logic checks).
DevTables bind_host_pointers(const HostTables& ht);

}  // namespace tp

// TEST INFRASTRUCTURE ONLY.
//
// CPU restatement of the batch tag pipeline of Mumicanso8323/ImageCaptioner
// (Program.cs).  It exists to CHECK the HIP implementation: only tests/,
// __graft_entry__.smoke() and bench.py's cpu_baseline leg load it, and only as
// the checker / the reported CPU baseline — never as the product path.
//
// Parity pinning: the reference cannot be built here (.NET is absent), so this
// restatement is pinned by the reference's only test, the eleven assertions of
// RunSelfChecks (Program.cs:1951-1983), reproduced in tests/test_oracle_selfcheck.py,
// and by hand-derived vectors for the shipped rules/*.json.  The corpus tag
// histogram has no counterpart in the reference (it only reads top_tags.txt,
// Program.cs:983-1034): that part is "parity unpinned" and build-defined.
//
// The restatement works on UTF-16 code units, as the reference's strings do,
// and follows the reference function by function; each function cites the
// lines it restates.  Known approximations (all outside ASCII): case mapping
// covers the BMP simple mappings only; culture-sensitive StartsWith
// (Program.cs:1126, 1191) is treated as ordinal.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <memory>
#include <sstream>
#include <string>
#include <thread>
#include <unordered_map>
#include <unordered_set>
#include <vector>

#include "../include/tagpipe.h"
#include "oracle_json.hpp"

namespace orc {

typedef char16_t C;

// ---------------------------------------------------------------------------------------------
// .NET string primitives
// ---------------------------------------------------------------------------------------------
#include "unicode_tables.inc"

static C map_lookup(const unsigned short (*tab)[2], int n, C c) {
    int lo = 0, hi = n - 1;
    while (lo <= hi) {
        int mid = (lo + hi) >> 1;
        if (tab[mid][0] == c) return (C)tab[mid][1];
        if (tab[mid][0] < c) lo = mid + 1; else hi = mid - 1;
    }
    return c;
}
// char.ToLowerInvariant (U+0130 maps to itself on .NET).
static inline C lower_inv(C c) {
    if (c < 0x80) return (c >= 'A' && c <= 'Z') ? (C)(c + 32) : c;
    return map_lookup(kLowerMap, kLowerMapCount, c);
}
// Ordinal upper-casing used by StringComparer.OrdinalIgnoreCase (U+0131 and U+017F map to themselves).
static inline C upper_ord(C c) {
    if (c < 0x80) return (c >= 'a' && c <= 'z') ? (C)(c - 32) : c;
    return map_lookup(kUpperMap, kUpperMapCount, c);
}
// char.IsWhiteSpace; the regex class \s uses the same predicate.
static inline bool is_ws(C c) {
    return c == 0x20 || (c >= 0x09 && c <= 0x0D) || c == 0x85 || c == 0xA0 || c == 0x1680 || (c >= 0x2000 && c <= 0x200A) ||
           c == 0x2028 || c == 0x2029 || c == 0x202F || c == 0x205F || c == 0x3000;
}
static S to_lower(const S& s) { S r(s); for (auto& c : r) c = lower_inv(c); return r; }
static S fold(const S& s) { S r(s); for (auto& c : r) c = upper_ord(c); return r; }
static bool is_blank(const S& s) { for (C c : s) if (!is_ws(c)) return false; return true; }  // string.IsNullOrWhiteSpace
static S trim_ws(const S& s) {
    size_t a = 0, b = s.size();
    while (a < b && is_ws(s[a])) ++a;
    while (b > a && is_ws(s[b - 1])) --b;
    return s.substr(a, b - a);
}
static S u(const char* ascii) { S r; for (; *ascii; ++ascii) r.push_back((C)(unsigned char)*ascii); return r; }
static bool contains(const S& s, const S& needle) { return s.find(needle) != S::npos; }
static bool starts_with(const S& s, const S& p) { return s.size() >= p.size() && s.compare(0, p.size(), p) == 0; }
static bool starts_with_oic(const S& s, const S& p) { return s.size() >= p.size() && fold(s.substr(0, p.size())) == fold(p); }
static size_t index_of_oic(const S& s, const S& p) {
    S fs = fold(s), fp = fold(p);
    return fs.find(fp);
}

static S utf8_to_16(const uint8_t* p, size_t n) {
    S out;
    size_t i = 0;
    while (i < n) {
        uint8_t c = p[i];
        if (c < 0x80) { out.push_back(c); ++i; continue; }
        int len = (c >= 0xF0 && c <= 0xF4) ? 4 : (c >= 0xE0) ? 3 : (c >= 0xC2) ? 2 : 0;
        bool ok = len && i + len <= n;
        uint32_t cp = 0;
        if (ok) {
            cp = (len == 2) ? (c & 0x1Fu) : (len == 3) ? (c & 0x0Fu) : (c & 0x07u);
            for (int k = 1; k < len; ++k) {
                if ((p[i + k] & 0xC0) != 0x80) { ok = false; break; }
                cp = (cp << 6) | (p[i + k] & 0x3Fu);
            }
            if (ok && ((len == 3 && cp < 0x800) || (len == 4 && (cp < 0x10000 || cp > 0x10FFFF)) || (cp >= 0xD800 && cp <= 0xDFFF))) ok = false;
        }
        if (!ok) { out.push_back(0xFFFD); ++i; continue; }
        if (cp >= 0x10000) { cp -= 0x10000; out.push_back((C)(0xD800 + (cp >> 10))); out.push_back((C)(0xDC00 + (cp & 0x3FF))); }
        else out.push_back((C)cp);
        i += len;
    }
    return out;
}
static void utf16_to_8(const S& s, std::string& out) {
    for (size_t i = 0; i < s.size(); ++i) {
        uint32_t cp = s[i];
        if (cp >= 0xD800 && cp <= 0xDBFF && i + 1 < s.size() && s[i + 1] >= 0xDC00 && s[i + 1] <= 0xDFFF) {
            cp = 0x10000 + ((cp - 0xD800) << 10) + (s[i + 1] - 0xDC00);
            ++i;
        } else if (cp >= 0xD800 && cp <= 0xDFFF) {
            cp = 0xFFFD;
        }
        if (cp < 0x80) out.push_back((char)cp);
        else if (cp < 0x800) { out.push_back((char)(0xC0 | (cp >> 6))); out.push_back((char)(0x80 | (cp & 0x3F))); }
        else if (cp < 0x10000) { out.push_back((char)(0xE0 | (cp >> 12))); out.push_back((char)(0x80 | ((cp >> 6) & 0x3F))); out.push_back((char)(0x80 | (cp & 0x3F))); }
        else { out.push_back((char)(0xF0 | (cp >> 18))); out.push_back((char)(0x80 | ((cp >> 12) & 0x3F))); out.push_back((char)(0x80 | ((cp >> 6) & 0x3F))); out.push_back((char)(0x80 | (cp & 0x3F))); }
    }
}
static std::string to8(const S& s) { std::string o; utf16_to_8(s, o); return o; }

// Dictionary<string, V>(StringComparer.OrdinalIgnoreCase): enumeration is insertion order (nothing is ever removed),
// and `d[k] = v` on an existing key keeps the key's position.
template <class V>
struct IMap {
    std::vector<std::pair<S, V>> items;
    std::unordered_map<S, size_t> idx;
    V* find(const S& k) { auto it = idx.find(fold(k)); return it == idx.end() ? nullptr : &items[it->second].second; }
    const V* find(const S& k) const { auto it = idx.find(fold(k)); return it == idx.end() ? nullptr : &items[it->second].second; }
    void set(const S& k, const V& v) {
        S f = fold(k);
        auto it = idx.find(f);
        if (it == idx.end()) { idx.emplace(f, items.size()); items.emplace_back(k, v); }
        else items[it->second].second = v;
    }
    size_t size() const { return items.size(); }
};
// HashSet<string>(StringComparer.OrdinalIgnoreCase) with insertion-order enumeration.
struct ISet {
    std::vector<S> items;
    std::unordered_set<S> keys;
    bool add(const S& s) { if (!keys.insert(fold(s)).second) return false; items.push_back(s); return true; }
    bool has(const S& s) const { return keys.count(fold(s)) != 0; }
    size_t size() const { return items.size(); }
    void clear() { items.clear(); keys.clear(); }
};
static ISet make_set(std::initializer_list<const char*> l) { ISet s; for (auto p : l) s.add(u(p)); return s; }
// Enumerable.Distinct(StringComparer.OrdinalIgnoreCase): first occurrences, in order.
static std::vector<S> distinct_oic(const std::vector<S>& v) { ISet s; for (auto& x : v) s.add(x); return s.items; }
// OrderBy(t => t, StringComparer.OrdinalIgnoreCase): stable, compares upper-cased code units.
static void sort_oic(std::vector<S>& v) {
    std::stable_sort(v.begin(), v.end(), [](const S& a, const S& b) { return fold(a) < fold(b); });
}

// ---------------------------------------------------------------------------------------------
// Static tables (Program.cs:27-126)
// ---------------------------------------------------------------------------------------------
enum Bucket { Identity, FaceHairEyes, Outfit, PoseAction, Background, StyleQuality, Other };  // Program.cs:128

struct ActionSubtype { S tag; std::vector<S> matchAny, matchAll; };                       // Program.cs:130
struct ActionRule { S parent; std::vector<S> reqAll, reqAny, forbAny; std::vector<ActionSubtype> subs; };  // :129
struct FacetRule { S tag; std::vector<S> matchAny, matchAll; };                           // :131
struct Facet { S name, def; std::vector<FacetRule> rules; };                              // :132
struct Lexicon { IMap<S> aliases; IMap<std::vector<S>> implied; };                        // :133

struct Candidate { S tag; double score; int order; int source; };                         // :391-396
struct Selection { std::vector<S> tags, droppedMale, droppedReasons; };                  // :403-407
struct Coverage { unsigned bits; std::vector<S> missingRequired, missingForPrompt, details; };  // :434-440

static std::vector<S> words(const char* bar_separated) {
    std::vector<S> out;
    std::string cur;
    for (const char* p = bar_separated;; ++p) {
        if (*p == '|' || *p == 0) { out.push_back(u(cur.c_str())); cur.clear(); if (!*p) break; }
        else cur.push_back(*p);
    }
    return out;
}
// The alternations of Program.cs:48-54, as literal lists ("contains any of", IgnoreCase).
static const std::vector<S> kFaceHairEyes = words("hair|eyes|eyebrows|eyelashes|pupil|bangs|twintails|ponytail|braid|ahoge|blush|smile|frown|angry|surprised|open_mouth|closed_mouth|tears|looking_");
static const std::vector<S> kOutfit = words("dress|skirt|shirt|jacket|coat|hoodie|sweater|uniform|kimono|swimsuit|socks|stockings|pantyhose|shoes|boots|gloves|hat|ribbon|bow|tie|neckwear|blazer|pants|shorts");
static const std::vector<S> kPose = words("pose|standing|sitting|lying|kneeling|crouching|walking|running|jumping|leaning|holding|raising|arms|hands|fingers|looking_at_viewer|looking_away|from_side|from_behind");
static const std::vector<S> kBackground = words("indoors|outdoors|room|bedroom|street|city|school|classroom|park|forest|beach|sky|night|sunset|window|bed|chair|office|kitchen|bathroom");
static const std::vector<S> kStyle = words("lighting|backlighting|rim_light|soft_light|hard_light|shadow|bloom|depth_of_field|bokeh|monochrome|highres|absurdres|masterpiece|best_quality|high_quality|detailed");
static const std::vector<S> kHairCoverage = words("hair|bangs|twintails|ponytail|braid|ahoge");
static const std::vector<S> kEyeCoverage = words("eyes|eyelashes|eyebrows|pupil|heterochromia");
static bool regex_any(const std::vector<S>& lits, const S& tag) {
    S t = to_lower(tag);
    for (auto& l : lits) if (contains(t, l)) return true;
    return false;
}
// Program.cs:56-58 (the tenth entry holds U+2019).
static std::vector<S> make_drop_contains() {
    std::vector<S> v = {u("negative:"), u("positive:"), u("sorry"), u("i can't"), u("i cannot"), u("cannot provide"), u("unable to"), u("as an ai"), u("i'm not able")};
    S w = u("i won"); w.push_back(0x2019); w.push_back('t');
    v.push_back(w);
    v.push_back(u("i won't"));
    return v;
}
static const std::vector<S> kDropTagsContains = make_drop_contains();
// Program.cs:43-45, declaration order (tp_verdict.safe_hit indexes this list).
static const char* const kSafeModeRejectList[20] = {"nsfw", "nude", "naked", "nipples", "breasts", "pussy", "cameltoe", "sex", "penis", "cum", "orgasm", "masturbation", "fellatio", "anal", "vaginal", "hentai", "explicit", "porn", "bdsm", "futanari"};

static const int MaxTotalTags = 42;      // Program.cs:38
static const int DefaultStyleTagCap = 2; // :400
static const int MaxStyleTagCap = 4;     // :401

struct FileDetail {
    std::vector<S> fired, dropped, droppedMale, unknown, evidence, finalTags, covDetails, captionTags;
    S stopReason, rejectReason;
    std::vector<S> coverageTrace;
    // BuildReviewRecord's result (Program.cs:826-868); present == false: it returned null
    struct Review { bool present = false; std::vector<S> reasons, banned, dropped, top, unknown; int finalTagCount = 0; bool shortCaption = false; } review;
};

// ---------------------------------------------------------------------------------------------
// Engine: the static state of Program plus the functions of the pure span
// ---------------------------------------------------------------------------------------------
struct Engine {
    // Program.cs:20-45, 60-78
    std::vector<S> HardPrefixTags;
    ISet AllowlistTags, DenylistTags, KnownTags;
    bool KnownTagDictionaryFound = false;
    ISet SubjectTokens = make_set({"1girl", "1boy", "solo", "multiple_girls", "multiple_boys", "2girls", "2boys", "girl", "boy"});
    ISet MaleSubjectTokens = make_set({"boy", "1boy", "2boys", "3boys", "4boys", "multiple_boys", "male_focus", "yaoi"});
    ISet BannedFinalTags = make_set({"1boy", "boy", "multiple_boys", "2boys", "3boys", "4boys", "male_focus", "yaoi", "male"});
    bool ActiveStrictKnownTags = false;
    ISet ForceAllowTags;
    bool hasProfile = false;
    std::vector<S> profileAlwaysAdd;
    ISet SafeModeRejectTags;
    ISet BuiltinDenyTags = make_set({"watermark", "signature", "text", "logo", "jpeg_artifacts", "lowres", "worst_quality", "rating", "artist_name"});
    ISet ActionTags = make_set({"footjob", "penis_between_feet", "foot_on_penis", "foot_sole_stimulation", "trampling_penis", "handjob", "penis_in_hand", "hand_wrapping_penis", "squeezing_penis", "sex", "vaginal_penetration", "missionary", "doggystyle", "cowgirl_position", "blowjob", "penis_in_mouth", "fellatio"});
    IMap<S> ActionTagAliases;
    std::vector<ActionRule> ActionInferenceRules;
    Lexicon ActiveLexicon;
    std::vector<ActionRule> ActiveActionRules;
    std::vector<Facet> ActiveRequiredFacets;

    // options
    S Trigger;
    S minimalFallback;
    bool strictExplicit = false, strictOpt = false, safeMode = true;
    double lineWeight = 1.0;
    int lineSource = TP_SRC_OPENAI;
    std::string newline = "\n";

    Engine() {
        for (auto p : kSafeModeRejectList) SafeModeRejectTags.add(u(p));
        // Program.cs:71-78
        ActionTagAliases.set(u("pedal_stimulation"), u("footjob"));
        ActionTagAliases.set(u("feet_stimulation"), u("footjob"));
        ActionTagAliases.set(u("foot_stimulation"), u("footjob"));
        ActionTagAliases.set(u("sexual_intercourse"), u("sex"));
        ActionTagAliases.set(u("paizuri_sex"), u("sex"));
        ActionTagAliases.set(u("fellatio"), u("blowjob"));
        // Program.cs:80-126
        auto L = [](std::initializer_list<const char*> l) { std::vector<S> v; for (auto p : l) v.push_back(u(p)); return v; };
        auto ST = [&](const char* tag, std::initializer_list<const char*> any) { ActionSubtype s; s.tag = u(tag); s.matchAny = L(any); return s; };
        ActionRule r1{u("footjob"), L({"feet", "penis"}), {}, {}, {ST("penis_between_feet", {"between_feet", "between", "sandwiched", "foot_sandwich"}), ST("foot_on_penis", {"foot_on_penis", "footjob", "foot_press", "foot_pressing"}), ST("foot_sole_stimulation", {"sole", "foot_sole", "sole_focus", "underside"}), ST("trampling_penis", {"stepping", "trampling", "step_on", "weight_on_penis"})}};
        ActionRule r2{u("handjob"), L({"penis", "hand"}), {}, L({"vaginal_penetration", "anal_penetration", "vaginal", "anal", "insertion", "penetration"}), {ST("penis_in_hand", {"penis_in_hand", "holding_penis", "holding"}), ST("hand_wrapping_penis", {"grip", "gripping", "hand_wrapping_penis", "wrapped_hand"}), ST("squeezing_penis", {"squeezing", "stroking", "jerking", "pumping_motion"})}};
        ActionRule r3{u("sex"), L({"penis"}), L({"vagina", "vaginal", "vaginal_penetration", "insertion", "penetration", "intercourse"}), {}, {ST("vaginal_penetration", {"vaginal_penetration", "vaginal", "penis_in_vagina", "insertion", "penetration"}), ST("missionary", {"missionary", "missionary_position", "face_to_face"}), ST("doggystyle", {"doggystyle", "doggy_style", "from_behind", "rear_entry"}), ST("cowgirl_position", {"cowgirl", "cowgirl_position", "female_on_top"})}};
        ActionRule r4{u("blowjob"), L({"penis"}), L({"mouth", "oral", "fellatio", "sucking"}), {}, {ST("penis_in_mouth", {"penis_in_mouth", "deepthroat", "mouth_penetration"}), ST("fellatio", {"fellatio", "oral", "oral_sex"})}};
        ActionInferenceRules = {r1, r2, r3, r4};
        ActiveLexicon = BuildDefaultLexicon();       // Program.cs:135
        ActiveActionRules = ActionInferenceRules;    // :136
        // ActiveRequiredFacets = []                 // :137
    }

    // ---- Program.cs:1267-1273 -------------------------------------------------------------
    static S NormalizeTagBasic(const S& in) {
        S t = trim_ws(in);                                        // t.Trim()
        {                                                         // .Trim('"', '\'', '`')
            size_t a = 0, b = t.size();
            auto q = [](C c) { return c == '"' || c == '\'' || c == '`'; };
            while (a < b && q(t[a])) ++a;
            while (b > a && q(t[b - 1])) --b;
            t = t.substr(a, b - a);
        }
        {   // Regex.Replace(t, @"[;。、\.]+$", ""): `$` also matches just before a final '\n'; the leftmost match is replaced.
            auto punct = [](C c) { return c == ';' || c == 0x3002 || c == 0x3001 || c == '.'; };
            size_t e = t.size();
            if (e > 0 && t[e - 1] == '\n') {
                size_t p = e - 1, q = p;
                while (q > 0 && punct(t[q - 1])) --q;
                t.erase(q, p - q);
            } else {
                while (e > 0 && punct(t[e - 1])) --e;
                t.resize(e);
            }
        }
        {   // \s+ -> "_", then _+ -> "_"
            S r;
            bool inws = false;
            for (C c : t) {
                if (is_ws(c)) { if (!inws) r.push_back('_'); inws = true; }
                else { r.push_back(c); inws = false; }
            }
            S r2;
            for (C c : r) if (!(c == '_' && !r2.empty() && r2.back() == '_')) r2.push_back(c);
            t = r2;
        }
        return to_lower(t);                                       // ToLowerInvariant
    }
    // ---- Program.cs:1149-1158 -------------------------------------------------------------
    S NormalizeTag(const S& in) const {
        S t = NormalizeTagBasic(in);
        if (const S* c = ActiveLexicon.aliases.find(t)) return *c;
        if (const S* c = ActionTagAliases.find(t)) return *c;
        return t;
    }
    // ---- Program.cs:1228-1253 -------------------------------------------------------------
    Lexicon BuildDefaultLexicon() const {
        Lexicon lex;
        for (auto& kv : ActionTagAliases.items) lex.aliases.set(NormalizeTagBasic(kv.first), NormalizeTagBasic(kv.second));
        lex.aliases.set(u("sole"), u("foot_sole"));
        lex.aliases.set(u("soles"), u("foot_sole"));
        lex.aliases.set(u("foot"), u("feet"));
        lex.aliases.set(u("hands"), u("hand"));
        lex.aliases.set(u("fingers"), u("hand"));
        lex.aliases.set(u("licking"), u("oral"));
        auto I = [&](const char* k, std::initializer_list<const char*> v) { std::vector<S> l; for (auto p : v) l.push_back(u(p)); lex.implied.set(u(k), l); };
        I("barefoot", {"feet"}); I("foot", {"feet"}); I("toes", {"feet"}); I("sole", {"feet"}); I("soles", {"feet"}); I("foot_sole", {"feet"});
        I("hands", {"hand"}); I("fingers", {"hand"}); I("oral", {"mouth"}); I("licking", {"oral", "mouth"});
        return lex;
    }
    // ---- Program.cs:1181-1187 -------------------------------------------------------------
    static std::vector<S> SplitTags(const S& s) {
        std::vector<S> out;
        if (is_blank(s)) return out;
        size_t i = 0;
        for (;;) {
            size_t j = s.find((C)',', i);
            S part = trim_ws(s.substr(i, (j == S::npos ? s.size() : j) - i));  // TrimEntries
            if (!part.empty()) {                                               // RemoveEmptyEntries
                S t = trim_ws(part);
                if (!is_blank(t)) out.push_back(t);
            }
            if (j == S::npos) break;
            i = j + 1;
        }
        return out;
    }
    // ---- Program.cs:1135-1139 -------------------------------------------------------------
    static bool IsValidTag(const S& tag) {
        if (is_blank(tag)) return false;
        if (tag[0] == ':') return false;
        // ^[a-z0-9][a-z0-9_()!'/-]*$  (case-sensitive; `$` would also accept a final '\n')
        size_t n = tag.size();
        if (n > 0 && tag[n - 1] == '\n') --n;
        if (n == 0) return false;
        auto first = [](C c) { return (c >= 'a' && c <= 'z') || (c >= '0' && c <= '9'); };
        auto rest = [&](C c) { return first(c) || c == '_' || c == '(' || c == ')' || c == '!' || c == '\'' || c == '/' || c == '-'; };
        if (!first(tag[0])) return false;
        for (size_t i = 1; i < n; ++i) if (!rest(tag[i])) return false;
        return true;
    }
    // ---- Program.cs:1121-1133 -------------------------------------------------------------
    bool ShouldDropTag(const S& t) const {
        S lower = to_lower(t);
        if (ForceAllowTags.has(lower)) return false;
        if (!IsValidTag(lower)) return true;
        for (auto& d : kDropTagsContains) if (contains(lower, d)) return true;
        if (starts_with(lower, u("rating:"))) return true;
        if (BuiltinDenyTags.has(lower)) return true;
        if (BannedFinalTags.has(lower)) return true;
        if (DenylistTags.has(lower)) return true;
        if (AllowlistTags.size() > 0 && !AllowlistTags.has(lower) && (starts_with(lower, u("rating")) || contains(lower, u("watermark")))) return true;
        if (KnownTagDictionaryFound && ActiveStrictKnownTags && !KnownTags.has(lower)) return true;
        return false;
    }
    // ---- Program.cs:1113-1119 -------------------------------------------------------------
    bool IsLikelyCharacterName(const S& tag) const {
        // ^[a-z0-9]+(_[a-z0-9]+){1,3}$ with IgnoreCase
        {
            size_t n = tag.size();
            if (n > 0 && tag[n - 1] == '\n') --n;
            auto alnum = [](C c) { return (c >= 'a' && c <= 'z') || (c >= 'A' && c <= 'Z') || (c >= '0' && c <= '9'); };
            int segs = 0;
            size_t i = 0;
            bool ok = true;
            while (i < n) {
                size_t j = i;
                while (j < n && alnum(tag[j])) ++j;
                if (j == i) { ok = false; break; }
                ++segs;
                if (j == n) { i = j; break; }
                if (tag[j] != '_') { ok = false; break; }
                i = j + 1;
                if (i == n) { ok = false; break; }
            }
            if (!ok || segs < 2 || segs > 4) return false;
        }
        if (SubjectTokens.has(tag)) return false;
        if (regex_any(kFaceHairEyes, tag) || regex_any(kOutfit, tag) || regex_any(kPose, tag) || regex_any(kBackground, tag) || regex_any(kStyle, tag)) return false;
        return true;
    }
    // ---- Program.cs:1103-1111 -------------------------------------------------------------
    Bucket ClassifyTag(const S& tag) const {
        if (SubjectTokens.has(tag) || IsLikelyCharacterName(tag)) return Identity;
        if (regex_any(kFaceHairEyes, tag)) return FaceHairEyes;
        if (regex_any(kOutfit, tag)) return Outfit;
        if (regex_any(kPose, tag) || ActionTags.has(tag)) return PoseAction;
        if (regex_any(kBackground, tag)) return Background;
        if (regex_any(kStyle, tag)) return StyleQuality;
        return Other;
    }
    // ---- Program.cs:1141-1147 -------------------------------------------------------------
    static bool LooksBad(const S& s) {
        if (is_blank(s)) return true;
        S lower = to_lower(s);
        for (auto& d : kDropTagsContains) if (contains(lower, d)) return true;
        size_t commas = 0;
        for (C c : s) if (c == ',') ++commas;
        return commas < 2 && s.size() > 40;
    }
    // ---- Program.cs:1189-1197 -------------------------------------------------------------
    static S StripCodeFences(const S& in) {
        S s = trim_ws(in);
        if (!starts_with(s, u("```"))) return s;
        size_t first = s.find((C)'\n');
        if (first != S::npos) s = s.substr(first + 1);
        size_t last = s.rfind(u("```"));
        if (last != S::npos) s = s.substr(0, last);
        return trim_ws(s);
    }
    // ---- Program.cs:1199-1205 -------------------------------------------------------------
    static S ExtractAfterPrefixLine(const S& text, const S& prefix) {
        size_t i = 0;
        for (;;) {
            size_t j = text.find((C)'\n', i);
            S t = trim_ws(text.substr(i, (j == S::npos ? text.size() : j) - i));
            if (starts_with_oic(t, prefix)) return trim_ws(t.substr(prefix.size()));
            if (j == S::npos) break;
            i = j + 1;
        }
        return S();
    }
    // ---- Program.cs:1207-1210 -------------------------------------------------------------
    static S CutAfterPrefix(const S& text, const S& prefix) {
        size_t idx = index_of_oic(text, prefix);
        return idx != S::npos ? trim_ws(text.substr(0, idx)) : text;
    }
    // ---- Program.cs:589-605 ---------------------------------------------------------------
    S NormalizeRawTagLine(const S& raw, bool* rejected = nullptr) const {
        if (rejected) *rejected = true;
        if (is_blank(raw)) return S();
        S s;
        for (size_t i = 0; i < raw.size(); ++i) {               // raw.Replace("\r\n", "\n")
            if (raw[i] == '\r' && i + 1 < raw.size() && raw[i + 1] == '\n') continue;
            s.push_back(raw[i]);
        }
        s = trim_ws(s);
        S pos = ExtractAfterPrefixLine(s, u("POSITIVE:"));
        if (!is_blank(pos)) s = pos;
        s = CutAfterPrefix(s, u("NEGATIVE:"));
        s = StripCodeFences(s);
        for (auto& c : s) if (c == '\n') c = ' ';
        s = trim_ws(s);
        if (LooksBad(s)) return S();
        if (rejected) *rejected = false;
        std::vector<S> tags;
        for (auto& part : SplitTags(s)) {
            S t = NormalizeTag(part);
            if (t.size() < 2) continue;
            if (ShouldDropTag(t)) continue;
            tags.push_back(t);
        }
        tags = distinct_oic(tags);
        S out;
        for (size_t i = 0; i < tags.size(); ++i) { if (i) out += u(", "); out += tags[i]; }
        return out;
    }
    // ---- Program.cs:574-587 ---------------------------------------------------------------
    std::vector<Candidate> ParseTagCandidatesFromLine(const S& raw, const double* defaultScore, double sourceWeight, int source, bool* rejected = nullptr) const {
        S norm = NormalizeRawTagLine(raw, rejected);
        std::vector<Candidate> list;
        int i = 0;
        for (auto& t : SplitTags(norm)) {
            double base = defaultScore ? *defaultScore : std::max(0.05, 1.0 - i * 0.01);
            list.push_back(Candidate{t, base * sourceWeight, i, source});
            ++i;
        }
        return list;
    }
    // ---- Program.cs:712-717 ---------------------------------------------------------------
    static int SourceRank(int source) { return source == 0 ? 0 : source == 1 ? 1 : source == 2 ? 2 : 3; }
    // ---- Program.cs:796-799 ---------------------------------------------------------------
    static int PreferSource(int a, int b, double scoreA, double scoreB) {
        if (std::fabs(scoreA - scoreB) > 0.03) return scoreA >= scoreB ? a : b;
        return SourceRank(a) <= SourceRank(b) ? a : b;
    }
    // ---- Program.cs:442-461 ---------------------------------------------------------------
    static std::vector<Candidate> MergeCandidateTags(const std::vector<Candidate>& a, const std::vector<Candidate>& b) {
        IMap<Candidate> map;
        auto feed = [&](const Candidate& c) {
            Candidate* old = map.find(c.tag);
            if (!old) { map.set(c.tag, c); return; }
            const Candidate& bestScore = c.score > old->score ? c : *old;
            int earliestOrder = std::min(old->order, c.order);
            bool close = std::fabs(old->score - c.score) <= 0.03;
            int source = PreferSource(old->source, c.source, close ? old->score : bestScore.score, close ? c.score : bestScore.score);
            Candidate merged{bestScore.tag, std::max(old->score, c.score), earliestOrder, source};
            *old = merged;
        };
        for (auto& c : a) feed(c);
        for (auto& c : b) feed(c);
        std::vector<Candidate> out;
        for (auto& kv : map.items) out.push_back(kv.second);
        std::stable_sort(out.begin(), out.end(), [](const Candidate& x, const Candidate& y) {
            if (x.score != y.score) return x.score > y.score;
            int rx = SourceRank(x.source), ry = SourceRank(y.source);
            if (rx != ry) return rx < ry;
            return x.order < y.order;
        });
        return out;
    }
    // ---- Program.cs:801-815 ---------------------------------------------------------------
    static std::vector<Candidate> LimitSecondaryTags(const std::vector<Candidate>& primary, const std::vector<Candidate>& secondary, int cap) {
        std::vector<Candidate> accepted;
        if (cap <= 0 || secondary.empty()) return accepted;
        ISet primarySet;
        for (auto& c : primary) primarySet.add(c.tag);
        std::vector<Candidate> sorted = secondary;
        std::stable_sort(sorted.begin(), sorted.end(), [](const Candidate& x, const Candidate& y) {
            if (x.score != y.score) return x.score > y.score;
            return x.order < y.order;
        });
        for (auto& c : sorted) {
            if (primarySet.has(c.tag)) { accepted.push_back(c); continue; }
            if (cap == 0) continue;
            accepted.push_back(c);
            cap--;
        }
        return accepted;
    }
    // ---- Program.cs:702-710 ---------------------------------------------------------------
    int DetermineStyleQualityCap(const std::vector<Candidate>& candidates) const {
        ISet strong;
        for (auto& c : candidates)
            if (ClassifyTag(c.tag) == StyleQuality && c.score >= 0.60) strong.add(c.tag);
        int n = (int)strong.size();
        return std::max(DefaultStyleTagCap, std::min(MaxStyleTagCap, n));
    }
    // ---- Program.cs:607-700 ---------------------------------------------------------------
    Selection SelectTags(const std::vector<Candidate>& candidates) const {
        int caps[6] = {3, 10, 8, 6, 4, DetermineStyleQualityCap(candidates)};
        Selection sel;
        {
            std::vector<S> dm;
            for (auto& c : candidates) { S t = NormalizeTag(c.tag); if (!is_blank(t) && BannedFinalTags.has(t)) dm.push_back(t); }
            dm = distinct_oic(dm);
            sort_oic(dm);
            sel.droppedMale = dm;
        }
        std::vector<S> droppedReasons;
        IMap<Candidate> unique;
        for (auto& c : candidates) {
            S n = NormalizeTag(c.tag);
            if (is_blank(n)) continue;
            if (!IsValidTag(n)) { droppedReasons.push_back(n + u(":invalid_tag")); continue; }
            if (ShouldDropTag(n)) { droppedReasons.push_back(n + u(":policy_drop")); continue; }
            Candidate normalized{n, c.score, c.order, c.source};
            Candidate* old = unique.find(n);
            if (!old) { unique.set(n, normalized); continue; }
            const Candidate* prefer = normalized.score > old->score ? &normalized : old;
            if (std::fabs(normalized.score - old->score) <= 0.03)
                prefer = SourceRank(normalized.source) < SourceRank(old->source) ? &normalized : old;
            Candidate merged{prefer->tag, std::max(normalized.score, old->score), std::min(normalized.order, old->order), prefer->source};
            *old = merged;
        }
        std::vector<Candidate> sorted;
        for (auto& kv : unique.items) sorted.push_back(kv.second);
        std::stable_sort(sorted.begin(), sorted.end(), [](const Candidate& a, const Candidate& b) {
            if (a.score != b.score) return a.score > b.score;
            int ra = SourceRank(a.source), rb = SourceRank(b.source);
            if (ra != rb) return ra < rb;
            return a.order < b.order;
        });
        std::vector<std::vector<const Candidate*>> grouped(7);
        for (auto& c : sorted) grouped[ClassifyTag(c.tag)].push_back(&c);

        std::vector<S>& result = sel.tags;
        ISet selected;
        auto AddOne = [&](const S& tag) -> bool {
            S n = NormalizeTag(tag);
            if (is_blank(n) || (int)result.size() >= MaxTotalTags) return false;
            if (!IsValidTag(n)) return false;
            if (ShouldDropTag(n) || !selected.add(n)) return false;
            result.push_back(n);
            return true;
        };
        auto FillBucket = [&](int bucket, int cap) {
            int used = 0;
            for (auto* c : grouped[bucket]) {
                if ((int)result.size() >= MaxTotalTags || used >= cap) break;
                if (AddOne(c->tag)) ++used;
            }
        };
        AddOne(Trigger);
        AddOne(u("1girl"));
        if (hasProfile) for (auto& t : profileAlwaysAdd) AddOne(NormalizeTag(t));
        for (int b = Identity; b <= StyleQuality; ++b) FillBucket(b, caps[b]);
        for (auto& c : sorted) {
            if ((int)result.size() >= MaxTotalTags) break;
            AddOne(c.tag);
        }
        sel.droppedReasons = distinct_oic(droppedReasons);
        return sel;
    }
    // ---- Program.cs:1056-1095 -------------------------------------------------------------
    Coverage EvaluateCoverage(const std::vector<S>& tags, const S& trigger) const {
        std::vector<S> nt;
        for (auto& t : tags) { S n = NormalizeTag(t); if (!is_blank(n)) nt.push_back(n); }
        nt = distinct_oic(nt);
        auto list_has = [&](const S& x) { for (auto& t : nt) if (t == x) return true; return false; };  // List.Contains: ordinal
        bool hasTrigger = list_has(NormalizeTag(trigger));
        bool has1Girl = list_has(u("1girl"));
        bool hasHair = false, hasEyes = false, hasOutfit = false, hasPose = false, hasBackground = false;
        for (auto& t : nt) {
            if (regex_any(kHairCoverage, t)) hasHair = true;
            if (regex_any(kEyeCoverage, t)) hasEyes = true;
            Bucket b = ClassifyTag(t);
            if (b == Outfit) hasOutfit = true;
            if (b == PoseAction) hasPose = true;
            if (b == Background) hasBackground = true;
        }
        Coverage cv;
        cv.bits = ((hasTrigger && has1Girl) ? TP_COV_IDENTITY : 0) | (hasHair ? TP_COV_HAIR : 0) | (hasEyes ? TP_COV_EYES : 0) |
                  (hasOutfit ? TP_COV_OUTFIT : 0) | (hasPose ? TP_COV_POSE_ACTION : 0) | (hasBackground ? TP_COV_BACKGROUND : 0);
        if (!hasTrigger || !has1Girl) cv.missingRequired.push_back(u("identity"));
        if (!hasHair) cv.missingRequired.push_back(u("hair"));
        if (!hasEyes) cv.missingRequired.push_back(u("eyes"));
        if (!hasOutfit) cv.missingRequired.push_back(u("outfit"));
        if (!hasPose) cv.missingRequired.push_back(u("pose_action"));
        cv.missingForPrompt = cv.missingRequired;
        if (!hasBackground) cv.missingForPrompt.push_back(u("background_scene"));
        auto okm = [](bool b) { return b ? u("ok") : u("missing"); };
        cv.details = {u("identity=") + okm(hasTrigger && has1Girl), u("hair=") + okm(hasHair), u("eyes=") + okm(hasEyes),
                      u("outfit=") + okm(hasOutfit), u("pose_action=") + okm(hasPose), u("background_scene=") + okm(hasBackground)};
        return cv;
    }
    // ---- Program.cs:1097-1101 -------------------------------------------------------------
    bool MissingCoreBuckets(const std::vector<S>& tags) const {
        bool hasFace = false, hasOutfit = false;
        for (auto& t : tags) { Bucket b = ClassifyTag(t); if (b == FaceHairEyes) hasFace = true; if (b == Outfit) hasOutfit = true; }
        return !hasFace || !hasOutfit;
    }
    // ---- Program.cs:721-729 ---------------------------------------------------------------
    ISet BuildEvidenceTags(const std::vector<Candidate>& raw) const {
        ISet set;
        for (auto& c : raw) {
            if (c.score < 0.28) continue;
            S n = NormalizeTag(c.tag);
            if (!is_blank(n)) set.add(n);
        }
        return set;
    }
    // ---- Program.cs:1160-1179 -------------------------------------------------------------
    void ExpandImpliedTags(ISet& tags) const {
        bool changed;
        do {
            changed = false;
            std::vector<S> snapshot;
            for (auto& t : tags.items) { S n = NormalizeTag(t); if (!is_blank(n)) snapshot.push_back(n); }
            snapshot = distinct_oic(snapshot);
            for (auto& tag : snapshot) {
                if (!tags.has(tag)) { tags.add(tag); changed = true; }
                if (const std::vector<S>* impliedList = ActiveLexicon.implied.find(tag)) {
                    for (auto& raw : *impliedList) {
                        S implied = NormalizeTag(raw);
                        if (is_blank(implied)) continue;
                        if (tags.add(implied)) changed = true;
                    }
                }
            }
        } while (changed);
    }
    // ---- Program.cs:777-794 ---------------------------------------------------------------
    static bool all_in(const ISet& set, const std::vector<S>& v) { for (auto& x : v) if (!set.has(x)) return false; return true; }
    static bool any_in(const ISet& set, const std::vector<S>& v) { for (auto& x : v) if (set.has(x)) return true; return false; }
    static bool MatchesActionRule(const ISet& e, const ActionRule& r) {
        if (!r.reqAll.empty() && !all_in(e, r.reqAll)) return false;
        if (!r.reqAny.empty() && !any_in(e, r.reqAny)) return false;
        if (any_in(e, r.forbAny)) return false;
        return true;
    }
    static bool MatchesAnyAll(const ISet& e, const std::vector<S>& any, const std::vector<S>& all) {
        if (!any.empty() && !any_in(e, any)) return false;
        if (!all.empty() && !all_in(e, all)) return false;
        return true;
    }
    // ---- Program.cs:731-753 ---------------------------------------------------------------
    std::vector<S> InferActionTags(const std::vector<S>& finalTags, const ISet& evidence, std::vector<S>& fired, std::vector<uint32_t>* firedBits) const {
        ISet finalSet;
        for (auto& t : finalTags) { S n = NormalizeTag(t); if (!is_blank(n)) finalSet.add(n); }
        std::vector<S> inferred = finalSet.items;
        size_t nRules = ActiveActionRules.size(), subBase = 0;
        for (size_t r = 0; r < nRules; ++r) {
            const ActionRule& rule = ActiveActionRules[r];
            size_t myBase = subBase;
            subBase += rule.subs.size();
            if (!MatchesActionRule(evidence, rule)) continue;
            if (finalSet.add(rule.parent)) inferred.push_back(rule.parent);
            fired.push_back(u("action:") + rule.parent);
            if (firedBits) (*firedBits)[r >> 5] |= 1u << (r & 31);
            for (size_t s = 0; s < rule.subs.size(); ++s) {
                const ActionSubtype& st = rule.subs[s];
                if (!MatchesAnyAll(evidence, st.matchAny, st.matchAll)) continue;
                if (finalSet.add(st.tag)) inferred.push_back(st.tag);
                fired.push_back(u("subtype:") + rule.parent + u(":") + st.tag);
                if (firedBits) { size_t bit = nRules + myBase + s; (*firedBits)[bit >> 5] |= 1u << (bit & 31); }
            }
        }
        return inferred;
    }
    // ---- Program.cs:755-775 ---------------------------------------------------------------
    void EnsureRequiredFacets(std::vector<S>& finalTags, const ISet& evidence, std::vector<S>& fired, uint16_t* choice) const {
        if (ActiveRequiredFacets.empty()) return;
        ISet finalSet;
        for (auto& t : finalTags) finalSet.add(t);
        for (size_t f = 0; f < ActiveRequiredFacets.size(); ++f) {
            const Facet& facet = ActiveRequiredFacets[f];
            S selected = facet.def;
            if (choice) choice[f] = 0xFFFF;
            for (size_t r = 0; r < facet.rules.size(); ++r) {
                const FacetRule& rule = facet.rules[r];
                if (!MatchesAnyAll(evidence, rule.matchAny, rule.matchAll)) continue;
                selected = rule.tag;
                fired.push_back(u("facet:") + facet.name + u(":") + rule.tag);
                if (choice) choice[f] = (uint16_t)r;
                break;
            }
            if (finalSet.add(selected)) finalTags.push_back(selected);
            if (selected == facet.def) fired.push_back(u("facet:") + facet.name + u(":") + facet.def + u(":default"));
        }
    }
    // ---- Program.cs:719 -------------------------------------------------------------------
    bool HasTag(const std::vector<S>& tags, const S& tag) const {
        S want = fold(NormalizeTag(tag));
        for (auto& t : tags) if (fold(NormalizeTag(t)) == want) return true;
        return false;
    }
    // ---- Program.cs:817-824 ---------------------------------------------------------------
    std::vector<S> GetUnknownTags(const std::vector<S>& tags) const {
        if (!KnownTagDictionaryFound || KnownTags.size() == 0) return {};
        std::vector<S> v;
        for (auto& t : tags) { S n = NormalizeTag(t); if (!is_blank(n) && !KnownTags.has(n)) v.push_back(n); }
        v = distinct_oic(v);
        sort_oic(v);
        return v;
    }
    // ---- Program.cs:971-974 ---------------------------------------------------------------
    std::vector<S> CaptionTags(const std::vector<S>& finalTags) const {
        std::vector<S> v;
        for (auto& t : finalTags) { S n = NormalizeTag(t); if (!is_blank(n)) v.push_back(n); }
        return distinct_oic(v);
    }

    // ---- Program.cs:826-868 (the fields that do not depend on paths and options) --------------
    // Regex ^\d+(girl|girls|boy|boys)$ with IgnoreCase (:837): \d is the Unicode category Nd (BMP runs from the generated table).
    static bool IsDecimalDigit(C c) {
        if (c >= '0' && c <= '9') return true;
        for (int k = 0; k < kDigitZeroCount; ++k) if (c >= kDigitZero[k] && c < kDigitZero[k] + 10) return true;
        return false;
    }
    static bool MultiPersonCount(const S& t) {
        size_t d = 0;
        while (d < t.size() && IsDecimalDigit(t[d])) ++d;
        if (d == 0) return false;
        S r = fold(t.substr(d));
        return r == fold(u("girl")) || r == fold(u("girls")) || r == fold(u("boy")) || r == fold(u("boys"));
    }
    FileDetail::Review BuildReviewRecord(const std::vector<Candidate>& rawCandidates, const Selection& selection, const std::vector<S>& unknownFinalTags,
                                         bool shortCaption, size_t traceCount, const S& coverageStopReason) const {
        std::vector<S> rawTokens;
        for (auto& c : rawCandidates) { S t = NormalizeTag(c.tag); if (!is_blank(t)) rawTokens.push_back(t); }
        std::vector<S> bannedDetected;
        for (auto& t : rawTokens) if (BannedFinalTags.has(t)) bannedDetected.push_back(t);
        bannedDetected = distinct_oic(bannedDetected);
        sort_oic(bannedDetected);
        std::vector<S> subjectTokens;
        for (auto& t : rawTokens) if (SubjectTokens.has(t) || contains(fold(t), fold(u("boys"))) || contains(fold(t), fold(u("girls")))) subjectTokens.push_back(t);
        subjectTokens = distinct_oic(subjectTokens);
        bool hasMale = !bannedDetected.empty(), hasGirl = false, hasMultiPerson = false;
        for (auto& t : subjectTokens) {
            if (MaleSubjectTokens.has(t)) hasMale = true;
            if (contains(fold(t), fold(u("girl")))) hasGirl = true;
            if (starts_with_oic(t, u("multiple_")) || MultiPersonCount(t)) hasMultiPerson = true;
        }
        FileDetail::Review r;
        if (hasMale) r.reasons.push_back(u("male_subject_token"));
        if (hasMale && hasGirl) r.reasons.push_back(u("mixed_subject_tokens"));
        if (hasMultiPerson) r.reasons.push_back(u("multi_person"));
        if (shortCaption) r.reasons.push_back(u("short_caption"));
        bool hasCoverageLog = traceCount > 1 || fold(coverageStopReason) != fold(u("coverage_satisfied"));
        if (r.reasons.empty() && bannedDetected.empty() && unknownFinalTags.empty() && !hasCoverageLog) return r;   // null
        r.present = true;
        r.reasons = distinct_oic(r.reasons);
        r.banned = bannedDetected;
        r.dropped = selection.droppedMale;
        r.finalTagCount = (int)selection.tags.size();
        r.shortCaption = shortCaption;
        std::vector<const Candidate*> byScore;
        for (auto& c : rawCandidates) byScore.push_back(&c);
        std::stable_sort(byScore.begin(), byScore.end(), [](const Candidate* a, const Candidate* b) { return a->score != b->score ? a->score > b->score : a->order < b->order; });
        for (size_t k = 0; k < byScore.size() && k < 20; ++k) r.top.push_back(byScore[k]->tag);
        r.unknown = unknownFinalTags;
        return r;
    }

    // ---- the pure span, Program.cs:223-229, 246-248, 292-329 (OpenAI branch off) ------------
    void run_candidates(const std::vector<Candidate>& finalCandidates, bool lineRejected, std::string& caption, tp_verdict& v,
                        FileDetail* det, uint32_t* firedBits, size_t firedWords, uint16_t* facetChoice, std::vector<S>* captionTagsOut) const {
        std::memset(&v, 0, sizeof v);
        v.safe_hit = -1;
        Selection finalSelection = SelectTags(finalCandidates);
        Coverage coverage = EvaluateCoverage(finalSelection.tags, Trigger);
        S stop = u("openai_unavailable_or_not_needed");
        if (!coverage.missingForPrompt.empty()) stop = u("openai_disabled_or_missing_api_key");               // :246-248
        if (coverage.missingForPrompt.empty() && stop == u("openai_unavailable_or_not_needed")) stop = u("coverage_satisfied");  // :292-294

        std::vector<S> finalTags = finalSelection.tags;
        bool fallback = false;
        if (finalTags.size() < 3) {                                                                           // :297-298
            finalTags = SelectTags(ParseTagCandidatesFromLine(minimalFallback, nullptr, 1.0, TP_SRC_OPENAI)).tags;
            fallback = true;
        }
        std::vector<S> fired;
        ISet evidence = BuildEvidenceTags(finalCandidates);                                                   // :301
        ExpandImpliedTags(evidence);                                                                          // :302
        ISet normalizedFinalSet;                                                                              // :303
        for (auto& t : finalTags) { S n = NormalizeTag(t); if (!is_blank(n)) normalizedFinalSet.add(n); }
        ExpandImpliedTags(normalizedFinalSet);                                                                // :304
        finalTags = normalizedFinalSet.items;                                                                 // :305
        std::vector<uint32_t> fb(firedWords ? firedWords : 1, 0);
        finalTags = InferActionTags(finalTags, evidence, fired, firedWords ? &fb : nullptr);                  // :306
        EnsureRequiredFacets(finalTags, evidence, fired, facetChoice);                                        // :307
        if (firedBits) for (size_t i = 0; i < firedWords; ++i) firedBits[i] = fb[i];

        bool eligible = HasTag(evidence.items, u("feet")) && HasTag(evidence.items, u("penis"));             // :309
        bool footjob = eligible && HasTag(finalTags, u("footjob"));                                          // :311
        std::vector<S> unknown = GetUnknownTags(finalTags);                                                   // :314
        std::vector<S> ctags = CaptionTags(finalTags);                                                        // :316
        S line;
        for (size_t i = 0; i < ctags.size(); ++i) { if (i) line += u(", "); line += ctags[i]; }
        caption.clear();
        utf16_to_8(line, caption);
        caption += newline;                                                                                   // :317
        bool isLow = MissingCoreBuckets(finalTags);                                                           // :319
        int safeHit = -1;
        S rejectReason;
        if (safeMode) {                                                                                       // :329, 942-945
            for (auto& t : finalTags) if (SafeModeRejectTags.has(t)) {
                S ft = fold(t);
                for (int k = 0; k < 20; ++k) if (fold(u(kSafeModeRejectList[k])) == ft) { safeHit = k; break; }
                rejectReason = u("safe_mode:") + t;
                break;
            }
        }
        auto clamp16 = [](size_t x) { return (uint16_t)std::min<size_t>(x, 0xFFFF); };
        v.coverage = (uint8_t)coverage.bits;
        v.flags = (uint8_t)((isLow ? TP_F_LOW : 0) | (eligible ? TP_F_ELIGIBLE : 0) | (footjob ? TP_F_FOOTJOB : 0) | (fallback ? TP_F_FALLBACK : 0) |
                            (lineRejected ? TP_F_LINE_REJECTED : 0) | (safeHit >= 0 ? TP_F_SAFE_REJECT : 0));
        v.n_tags = clamp16(ctags.size());
        v.n_candidates = clamp16(finalCandidates.size());
        v.n_selected = clamp16(finalSelection.tags.size());
        v.safe_hit = (int16_t)safeHit;
        v.n_evidence = clamp16(evidence.size());
        v.n_unknown = clamp16(unknown.size());
        v.n_dropped = clamp16(finalSelection.droppedReasons.size());
        if (captionTagsOut) *captionTagsOut = ctags;
        if (det) {
            det->fired = fired; det->dropped = finalSelection.droppedReasons; det->droppedMale = finalSelection.droppedMale;
            det->unknown = unknown; det->evidence = evidence.items; det->finalTags = finalTags; det->covDetails = coverage.details;
            det->captionTags = ctags; det->stopReason = stop; det->rejectReason = rejectReason;
            // coverageTrace's only entry when the OpenAI passes are off, Program.cs:228
            auto bar = [](const std::vector<S>& v) { S o; for (size_t i = 0; i < v.size(); ++i) { if (i) o += u("|"); o += v[i]; } return o; };
            const std::string cnt = std::to_string(finalSelection.tags.size());
            det->review = BuildReviewRecord(finalCandidates, finalSelection, unknown, isLow, /*traceCount=*/1, stop);   // :328
            det->coverageTrace = {u("pass0_joy missing=") + bar(coverage.missingForPrompt) + u(" required_missing=") + bar(coverage.missingRequired) +
                                  u(" details=") + bar(coverage.details) + u(" tags=") + u(cnt.c_str())};
        }
    }

    // =========================================================================================
    // Loaders
    // =========================================================================================
    static bool read_source(const tp_source& src, std::vector<uint8_t>& out) {
        out.clear();
        if (src.data) { out.assign(src.data, src.data + src.len); return true; }
        if (!src.path || !*src.path) return false;
        std::ifstream f(src.path, std::ios::binary);
        if (!f) return false;                                   // File.Exists == false
        out.assign(std::istreambuf_iterator<char>(f), std::istreambuf_iterator<char>());
        return true;
    }
    static S decode_text(const std::vector<uint8_t>& b) {
        size_t o = (b.size() >= 3 && b[0] == 0xEF && b[1] == 0xBB && b[2] == 0xBF) ? 3 : 0;
        return utf8_to_16(b.data() + o, b.size() - o);
    }
    // File.ReadLines: lines end at "\n", "\r" or "\r\n".
    static std::vector<S> read_lines(const S& text) {
        std::vector<S> lines;
        S cur;
        bool any = false;
        for (size_t i = 0; i < text.size(); ++i) {
            C c = text[i];
            if (c == '\r' || c == '\n') {
                if (c == '\r' && i + 1 < text.size() && text[i + 1] == '\n') ++i;
                lines.push_back(cur); cur.clear(); any = false;
            } else { cur.push_back(c); any = true; }
        }
        if (any) lines.push_back(cur);
        return lines;
    }
    // ---- Program.cs:1429-1439 -------------------------------------------------------------
    ISet LoadTagFile(const tp_source& src) const {
        ISet set;
        std::vector<uint8_t> b;
        if (!read_source(src, b)) return set;
        for (auto& line : read_lines(decode_text(b))) {
            S t = trim_ws(line);
            if (is_blank(t) || t[0] == '#') continue;
            set.add(NormalizeTag(t));
        }
        return set;
    }
    // ---- Program.cs:1036-1053 -------------------------------------------------------------
    void ExtractKnownTagsFromJson(const JV& el, ISet& tags) const {
        if (el.kind == JV::String) { S n = NormalizeTag(el.str); if (!is_blank(n)) tags.add(n); }
        else if (el.kind == JV::Array) { for (auto& c : el.arr) ExtractKnownTagsFromJson(c, tags); }
        else if (el.kind == JV::Object) {
            for (auto& p : el.obj) {
                ExtractKnownTagsFromJson(p.second, tags);
                S pn = NormalizeTag(p.first);
                if (!is_blank(pn)) tags.add(pn);
            }
        }
    }
    // ---- Program.cs:1013-1034 -------------------------------------------------------------
    bool TryLoadKnownTagsFromFile(const std::vector<uint8_t>& bytes, bool isJson, ISet& tags) const {
        tags.clear();
        try {
            if (isJson) {
                JsonParser jp(bytes.data(), bytes.size());
                JV doc = jp.parse_document();
                ExtractKnownTagsFromJson(doc, tags);
            } else {
                for (auto& line : read_lines(decode_text(bytes))) {
                    S cleaned = trim_ws(line);
                    if (is_blank(cleaned)) continue;
                    size_t i = 0;
                    for (;;) {
                        size_t j = cleaned.find_first_of(u(",;"), i);
                        S raw = trim_ws(cleaned.substr(i, (j == S::npos ? cleaned.size() : j) - i));
                        if (!raw.empty()) { S n = NormalizeTag(raw); if (!is_blank(n)) tags.add(n); }
                        if (j == S::npos) break;
                        i = j + 1;
                    }
                }
            }
            return tags.size() > 0;
        } catch (const JsonError&) {
            return false;
        }
    }
    // ---- Program.cs:983-1011 --------------------------------------------------------------
    void LoadKnownTagDictionary(const tp_config& cfg) {
        KnownTags.clear();
        KnownTagDictionaryFound = false;
        std::vector<uint8_t> b;
        if (read_source(cfg.vocabulary, b)) {
            bool isJson = cfg.vocabulary_is_json == 1;
            if (cfg.vocabulary_is_json < 0 && cfg.vocabulary.path) {
                std::string p(cfg.vocabulary.path);
                size_t dot = p.find_last_of('.');
                std::string ext = dot == std::string::npos ? "" : p.substr(dot);
                for (auto& ch : ext) ch = (char)std::tolower((unsigned char)ch);
                isJson = ext == ".json";
            }
            ISet loaded;
            if (TryLoadKnownTagsFromFile(b, isJson, loaded) && loaded.size() > 0) {
                KnownTags = loaded;
                KnownTagDictionaryFound = true;
                if (!strictExplicit) strictOpt = true;
                ActiveStrictKnownTags = strictOpt;
                return;
            }
        }
        if (!strictExplicit) strictOpt = false;
        ActiveStrictKnownTags = strictOpt;
    }
    // ---- Program.cs:918-931 ---------------------------------------------------------------
    // ProfileConfig is bound with PropertyNameCaseInsensitive only: JSON keys must equal
    // "Trigger" / "AlwaysAdd" / "PreferKeep" ignoring case ("always_add" does not bind).
    void LoadProfile(const tp_config& cfg) {
        hasProfile = false;
        profileAlwaysAdd.clear();
        std::vector<uint8_t> b;
        if (!read_source(cfg.profile, b)) return;
        bool triggerNull = false, alwaysNull = false, alwaysHasNull = false;
        S trig = u("sgrui");
        std::vector<S> always;
        try {
            JsonParser jp(b.data(), b.size());
            JV doc = jp.parse_document();
            if (doc.kind == JV::Null) return;
            if (doc.kind != JV::Object) return;               // JsonException -> catch -> null
            auto ieq = [](const S& a, const char* b2) { return fold(a) == fold(u(b2)); };
            auto string_list = [](const JV& v, std::vector<S>& out, bool& isNull, bool& hasNull) {
                out.clear(); isNull = false; hasNull = false;
                if (v.kind == JV::Null) { isNull = true; return; }
                if (v.kind != JV::Array) throw JsonError("cannot bind to List<string>");
                for (auto& e : v.arr) {
                    if (e.kind == JV::Null) { hasNull = true; out.push_back(S()); }
                    else if (e.kind == JV::String) out.push_back(e.str);
                    else throw JsonError("cannot bind to string");
                }
            };
            for (auto& p : doc.obj) {
                if (ieq(p.first, "Trigger")) {
                    if (p.second.kind == JV::Null) { triggerNull = true; }
                    else if (p.second.kind == JV::String) { trig = p.second.str; triggerNull = false; }
                    else throw JsonError("cannot bind to string");
                } else if (ieq(p.first, "AlwaysAdd")) {
                    string_list(p.second, always, alwaysNull, alwaysHasNull);
                } else if (ieq(p.first, "PreferKeep")) {
                    std::vector<S> tmp; bool a, c2;
                    string_list(p.second, tmp, a, c2);
                }
            }
        } catch (const JsonError&) {
            return;
        }
        if (!triggerNull && !is_blank(trig)) Trigger = NormalizeTag(trig);          // :925 (runs before :926 can throw)
        if (alwaysNull || alwaysHasNull) return;                                     // :926 throws -> catch -> null
        std::vector<S> aa;
        for (auto& t : always) { S n = NormalizeTag(t); if (!is_blank(n)) aa.push_back(n); }
        profileAlwaysAdd = distinct_oic(aa);
        hasProfile = true;
    }
    // ---- Program.cs:933-940 ---------------------------------------------------------------
    void BuildForceAllowTags() {
        ForceAllowTags.clear();
        ForceAllowTags.add(NormalizeTag(Trigger));
        ForceAllowTags.add(u("1girl"));
        if (hasProfile) for (auto& t : profileAlwaysAdd) ForceAllowTags.add(NormalizeTag(t));
    }
    // ---- Program.cs:1408-1415 -------------------------------------------------------------
    static std::vector<S> ParseStringArray(const JV& el) {
        if (el.kind == JV::String) { S s = NormalizeTagBasic(el.str); return is_blank(s) ? std::vector<S>{} : std::vector<S>{s}; }
        if (el.kind != JV::Array) return {};
        std::vector<S> v;
        for (auto& x : el.arr) { S t = NormalizeTagBasic(x.get_string_or_empty()); if (!is_blank(t)) v.push_back(t); }
        return distinct_oic(v);
    }
    // ---- Program.cs:1275-1305 -------------------------------------------------------------
    void LoadLexicon(const tp_source& src) {
        std::vector<uint8_t> b;
        if (!read_source(src, b)) return;
        JsonParser jp(b.data(), b.size());
        JV root = jp.parse_document();
        Lexicon lex = ActiveLexicon;
        const JV* el;
        if ((el = root.get("aliases")) && el->is_obj())
            for (auto& p : el->obj) lex.aliases.set(NormalizeTagBasic(p.first), NormalizeTagBasic(p.second.get_string_or_empty()));
        if ((el = root.get("normalization")) && el->is_obj())
            for (auto& p : el->obj) lex.aliases.set(NormalizeTagBasic(p.first), NormalizeTagBasic(p.second.get_string_or_empty()));
        if ((el = root.get("synonyms")) && el->is_obj())
            for (auto& p : el->obj) {
                if (!p.second.is_arr()) continue;
                S canonical = NormalizeTagBasic(p.first);
                for (auto& item : p.second.arr) lex.aliases.set(NormalizeTagBasic(item.get_string_or_empty()), canonical);
            }
        if ((el = root.get("implied")) && el->is_obj())
            for (auto& p : el->obj) lex.implied.set(NormalizeTagBasic(p.first), ParseStringArray(p.second));
        if ((el = root.get("implied_tags")) && el->is_obj())
            for (auto& p : el->obj) lex.implied.set(NormalizeTagBasic(p.first), ParseStringArray(p.second));
        ActiveLexicon = lex;
    }
    // ---- Program.cs:1340-1372 -------------------------------------------------------------
    static bool ParseActionRule(const JV& item, ActionRule& out) {
        if (!item.is_obj()) return false;
        S parent;
        bool haveParent = false;
        if (const JV* pt = item.get("parent_tag")) {
            if (pt->kind == JV::String) { parent = pt->str; haveParent = true; }
            else if (pt->kind != JV::Null) throw JsonError("GetString on a non-string element");
        }
        if (!haveParent || is_blank(parent)) {
            if (const JV* pe = item.get("parent")) {
                if (pe->kind == JV::String) { parent = pe->str; haveParent = true; }
                else if (pe->kind == JV::Array) {
                    haveParent = false;
                    for (auto& x : pe->arr) {
                        if (x.kind != JV::String && x.kind != JV::Null) throw JsonError("GetString on a non-string element");
                        if (x.kind == JV::String && !is_blank(x.str)) { parent = x.str; haveParent = true; break; }
                    }
                    if (!haveParent) parent.clear();
                }
            }
        }
        if (is_blank(parent)) return false;
        const JV* e;
        out.reqAll = (e = item.get("required_all")) ? ParseStringArray(*e) : std::vector<S>{};
        out.reqAny = (e = item.get("required_any")) ? ParseStringArray(*e) : std::vector<S>{};
        out.forbAny = (e = item.get("forbidden_any")) ? ParseStringArray(*e) : std::vector<S>{};
        out.subs.clear();
        if ((e = item.get("subtypes")) && e->is_arr()) {
            for (auto& st : e->arr) {
                if (st.kind == JV::String) {
                    S tag = NormalizeTagBasic(st.str);
                    if (!is_blank(tag)) out.subs.push_back(ActionSubtype{tag, {tag}, {}});
                } else if (st.is_obj()) {
                    const JV* te = st.get("tag");
                    S tag = te ? NormalizeTagBasic(te->get_string_or_empty()) : S();
                    if (is_blank(tag)) continue;
                    const JV* m;
                    std::vector<S> any = (m = st.get("match_any")) ? ParseStringArray(*m) : std::vector<S>{};
                    std::vector<S> all = (m = st.get("match_all")) ? ParseStringArray(*m) : std::vector<S>{};
                    out.subs.push_back(ActionSubtype{tag, any, all});
                }
            }
        }
        out.parent = NormalizeTagBasic(parent);
        return true;
    }
    // ---- Program.cs:1307-1338 -------------------------------------------------------------
    void LoadActions(const tp_source& src) {
        std::vector<uint8_t> b;
        if (!read_source(src, b)) return;
        JsonParser jp(b.data(), b.size());
        JV root = jp.parse_document();
        std::vector<ActionRule> list;
        if (root.is_arr()) {
            for (auto& item : root.arr) { ActionRule r; if (ParseActionRule(item, r)) list.push_back(r); }
        } else if (root.is_obj()) {
            bool hasRuleFields = root.get("parent_tag") || root.get("parent");
            if (hasRuleFields) {
                ActionRule r;
                if (ParseActionRule(root, r)) list.push_back(r);
            } else {
                for (auto& p : root.obj) {
                    const JV& category = p.second;
                    if (!category.is_obj()) continue;
                    const JV* e;
                    std::vector<S> parents = (e = category.get("parent")) ? ParseStringArray(*e) : std::vector<S>{};
                    std::vector<S> subtypes = (e = category.get("subtypes")) ? ParseStringArray(*e) : std::vector<S>{};
                    for (auto& parent : parents) {
                        if (is_blank(parent)) continue;
                        ActionRule r;
                        r.parent = parent;
                        r.reqAll = {parent};
                        for (auto& st : subtypes) r.subs.push_back(ActionSubtype{st, {st}, {}});
                        list.push_back(r);
                    }
                }
            }
        }
        if (!list.empty()) ActiveActionRules = list;
    }
    // ---- Program.cs:1255-1265 -------------------------------------------------------------
    static std::vector<Facet> BuildDefaultRequiredFacets() {
        auto L = [](std::initializer_list<const char*> l) { std::vector<S> v; for (auto p : l) v.push_back(u(p)); return v; };
        auto R = [&](const char* tag, std::initializer_list<const char*> any) { return FacetRule{u(tag), L(any), {}}; };
        return {
            Facet{u("viewpoint"), u("unknown_pov"), {R("male_pov", {"male_pov", "pov_male", "from_male_pov"}), R("third_person", {"third_person", "third_person_view"})}},
            Facet{u("gaze"), u("unknown_gaze"), {R("looking_at_camera", {"looking_at_viewer", "looking_at_camera"}), R("looking_at_partner", {"looking_at_male", "looking_at_partner"}), R("looking_at_object", {"looking_at_penis", "looking_at_object"})}},
            Facet{u("underwear"), u("unknown_underwear"), {R("panties_aside", {"panties_aside"}), R("panties_off", {"panties_off", "no_panties"}), R("panties_on", {"panties_on", "wearing_panties"})}},
            Facet{u("fluids"), u("unknown_fluids"), {R("semen_present", {"semen", "cum"}), R("precum_present", {"pre_ejaculate", "precum"}), R("no_fluids", {"no_fluids"})}},
            Facet{u("ejaculation"), u("unknown_ejaculation"), {R("ejaculation", {"ejaculation", "cumshot"}), R("no_ejaculation", {"no_ejaculation"})}},
            Facet{u("expression"), u("unknown_expression"), {R("ahegao", {"ahegao"}), R("smile", {"smile", "smiling"}), R("neutral", {"neutral_expression"})}},
            Facet{u("mouth_shape"), u("unknown_mouth"), {R("tongue_out", {"tongue_out"}), R("mouth_open", {"open_mouth"}), R("mouth_closed", {"closed_mouth"})}},
        };
    }
    // ---- Program.cs:1374-1406 -------------------------------------------------------------
    void LoadRequiredFacets(const tp_source& src) {
        std::vector<uint8_t> b;
        if (!read_source(src, b)) return;
        JsonParser jp(b.data(), b.size());
        JV root = jp.parse_document();
        std::vector<Facet> facets;
        if (!root.is_obj()) return;
        for (auto& p : root.obj) {
            if (p.second.is_arr()) {
                std::vector<S> tags = ParseStringArray(p.second);
                if (tags.empty()) continue;
                S def;
                bool found = false;
                for (auto& t : tags) if (starts_with_oic(t, u("unknown_"))) { def = t; found = true; break; }
                if (!found) def = u("unknown_") + NormalizeTagBasic(p.first);
                std::vector<FacetRule> rules;
                for (auto& t : tags) if (!starts_with_oic(t, u("unknown_"))) rules.push_back(FacetRule{t, {t}, {}});
                facets.push_back(Facet{NormalizeTagBasic(p.first), def, rules});
            } else if (p.second.is_obj()) {
                const JV* de = p.second.get("default");
                S def = de ? NormalizeTagBasic(de->get_string_or_empty()) : u("unknown_") + NormalizeTagBasic(p.first);
                std::vector<FacetRule> rules;
                const JV* re = p.second.get("rules");
                if (re && re->is_arr()) {
                    for (auto& r : re->arr) {
                        if (!r.is_obj()) continue;
                        const JV* te = r.get("tag");
                        S tag = te ? NormalizeTagBasic(te->get_string_or_empty()) : S();
                        if (is_blank(tag)) continue;
                        const JV* m;
                        std::vector<S> any = (m = r.get("match_any")) ? ParseStringArray(*m) : std::vector<S>{};
                        std::vector<S> all = (m = r.get("match_all")) ? ParseStringArray(*m) : std::vector<S>{};
                        rules.push_back(FacetRule{tag, any, all});
                    }
                }
                facets.push_back(Facet{NormalizeTagBasic(p.first), def, rules});
            }
        }
        if (!facets.empty()) ActiveRequiredFacets = facets;
    }
    // ---- Program.cs:1212-1226, 1417-1422 ---------------------------------------------------
    void LoadRuleEngineConfigs(const tp_config& cfg) {
        ActiveLexicon = BuildDefaultLexicon();
        ActiveActionRules = ActionInferenceRules;
        ActiveRequiredFacets = BuildDefaultRequiredFacets();
        LoadLexicon(cfg.lexicon);
        LoadActions(cfg.actions);
        LoadRequiredFacets(cfg.required_facets);
        for (auto& rule : ActiveActionRules) { ActionTags.add(rule.parent); for (auto& st : rule.subs) ActionTags.add(st.tag); }
    }
    // ---- Program.cs:156-165 ---------------------------------------------------------------
    void configure(const tp_config& cfg) {
        auto cstr = [](const char* p, const char* dflt) { return p ? std::string(p) : std::string(dflt); };
        std::string trig = cstr(cfg.trigger, "sgrui");
        Trigger = utf8_to_16((const uint8_t*)trig.data(), trig.size());
        strictExplicit = cfg.strict_known_explicit != 0;
        strictOpt = cfg.strict_known != 0;
        ActiveStrictKnownTags = strictOpt;                                                   // :156
        safeMode = cfg.safe_mode != 0;
        lineWeight = cfg.line_weight;
        lineSource = cfg.line_source;
        newline = cstr(cfg.newline, "\n");
        std::string ap = cstr(cfg.auto_prefix_tags, ""), ms = cstr(cfg.minimal_subject_tags, "");
        HardPrefixTags = distinct_oic(SplitTags(utf8_to_16((const uint8_t*)ap.data(), ap.size())));  // :158
        AllowlistTags = LoadTagFile(cfg.allowlist);                                          // :159
        DenylistTags = LoadTagFile(cfg.denylist);
        LoadKnownTagDictionary(cfg);                                                         // :160
        if (!cfg.no_profile) LoadProfile(cfg);                                               // :161
        BuildForceAllowTags();                                                               // :162
        LoadRuleEngineConfigs(cfg);                                                          // :163
        std::vector<S> mf = HardPrefixTags;                                                  // :165
        for (auto& t : SplitTags(utf8_to_16((const uint8_t*)ms.data(), ms.size()))) mf.push_back(t);
        mf = distinct_oic(mf);
        minimalFallback.clear();
        for (size_t i = 0; i < mf.size(); ++i) { if (i) minimalFallback += u(", "); minimalFallback += mf[i]; }
    }
};

// ---------------------------------------------------------------------------------------------
// Batch driver + C API
// ---------------------------------------------------------------------------------------------
struct Batch {
    std::vector<uint8_t> captions;
    std::vector<uint64_t> offsets;
    std::vector<tp_verdict> verdicts;
    std::vector<uint32_t> fired;
    std::vector<uint16_t> facetChoice;
    std::vector<FileDetail> details;
    std::unordered_map<std::string, uint64_t> hist;
    std::string histText, detailText;
};

struct Ctx {
    Engine eng;
    size_t nSubtypes = 0, firedWords = 0;
};

static thread_local std::string g_err;

// secondaryBegin != nullptr: tags [tagBegin[i], secondaryBegin[i]) are the primary tagger reply, [secondaryBegin[i], tagBegin[i+1]) the
// secondary one, merged as Program.cs:214-221 does (sources is not read then).
struct ScoredInput { const uint8_t* arena; const uint64_t* tagOffsets; const uint64_t* tagBegin; const double* scores; const uint8_t* sources;
                     const uint64_t* secondaryBegin; int secondaryCap;
                     // or the tagger's reply bodies themselves: replyOffsets != nullptr (secondary ones when secondaryReplyOffsets != nullptr)
                     const uint8_t* replies = nullptr; const uint64_t* replyOffsets = nullptr;
                     const uint8_t* secondaryReplies = nullptr; const uint64_t* secondaryReplyOffsets = nullptr; uint8_t* replyStatus = nullptr; };

static void run_batch(const Ctx& cx, const uint8_t* arena, const uint64_t* offsets, const ScoredInput* scored, uint64_t n, bool wantDetail, int threads,
                      Batch& b, tp_counters& ctr) {
    const Engine& e = cx.eng;
    size_t nf = e.ActiveRequiredFacets.size();
    std::vector<std::string> caps(n);
    b.verdicts.assign(n, tp_verdict{});
    b.fired.assign(n * cx.firedWords, 0);
    b.facetChoice.assign(n * nf, 0xFFFF);
    if (wantDetail) b.details.assign(n, FileDetail{});
    std::vector<std::vector<S>> ctags(n);
    auto work = [&](uint64_t lo, uint64_t hi) {
        for (uint64_t i = lo; i < hi; ++i) {
            std::vector<Candidate> cands;
            bool rejected = false;
            if (!scored) {
                S raw = utf8_to_16(arena + offsets[i], (size_t)(offsets[i + 1] - offsets[i]));
                cands = e.ParseTagCandidatesFromLine(raw, nullptr, e.lineWeight, e.lineSource, &rejected);
            } else if (scored->replyOffsets) {
                // CallJoyTagAsync from the reply body on (Program.cs:500-505) and ParseJoyTags (Program.cs:463-489).
                // status: 0 parsed, 1 no "tags" array (empty list), 3 the reference throws (the image fails; an empty list here)
                uint8_t status = 0;
                auto reply = [&](const uint8_t* p, size_t len, int source) {
                    std::vector<Candidate> list;
                    try {
                        JsonParser jp(p, len);
                        JV doc = jp.parse_document();
                        const JV* tagsEl = doc.get("tags");
                        if (!tagsEl || tagsEl->kind != JV::Array) { status = std::max<uint8_t>(status, 1); return list; }
                        bool hasScore = false;
                        int order = 0;
                        auto number = [](const JV& v) {                 // TryGetDouble: InvalidOperationException unless a number
                            if (v.kind != JV::Number) throw JsonError("TryGetDouble on an element that is not a number");
                            return std::strtod(to8(v.str).c_str(), nullptr);
                        };
                        for (const JV& t : tagsEl->arr) {
                            if (t.kind == JV::String) {
                                S sn = e.NormalizeTag(t.str);
                                if (!is_blank(sn)) { list.push_back(Candidate{sn, std::max(0.01, 1.0 - order * 0.001), order, source}); ++order; }
                                continue;
                            }
                            if (t.kind != JV::Object) continue;
                            S tag;
                            const JV* el;
                            if ((el = t.get("tag")) && el->kind == JV::String) tag = el->str;
                            else if ((el = t.get("name")) && el->kind == JV::String) tag = el->str;
                            double score = 0.5;
                            if ((el = t.get("score"))) { score = number(*el); hasScore = true; }
                            else if ((el = t.get("confidence"))) { score = number(*el); hasScore = true; }
                            S nn = e.NormalizeTag(tag);
                            if (!is_blank(nn)) { list.push_back(Candidate{nn, score, order, source}); ++order; }
                        }
                        if (hasScore)
                            std::stable_sort(list.begin(), list.end(), [](const Candidate& a, const Candidate& c) { return a.score != c.score ? a.score > c.score : a.order < c.order; });
                    } catch (const JsonError&) {
                        status = 3;
                        list.clear();
                    }
                    return list;
                };
                std::vector<Candidate> primary = reply(scored->replies + scored->replyOffsets[i], (size_t)(scored->replyOffsets[i + 1] - scored->replyOffsets[i]),
                                                       TP_SRC_PRIMARY_JOY);
                if (!scored->secondaryReplyOffsets) cands = primary;
                else {
                    std::vector<Candidate> secondary = reply(scored->secondaryReplies + scored->secondaryReplyOffsets[i],
                                                             (size_t)(scored->secondaryReplyOffsets[i + 1] - scored->secondaryReplyOffsets[i]), TP_SRC_SECONDARY_JOY);
                    cands = Engine::MergeCandidateTags(primary, Engine::LimitSecondaryTags(primary, secondary, scored->secondaryCap));
                }
                if (status == 3) cands.clear();
                if (scored->replyStatus) scored->replyStatus[i] = status;
            } else {
                // ParseJoyTags, object-shaped entries (Program.cs:473-487)
                auto parse = [&](uint64_t lo, uint64_t hi, int fixedSource) {
                    std::vector<Candidate> list;
                    int order = 0;
                    for (uint64_t t = lo; t < hi; ++t) {
                        S raw = utf8_to_16(scored->arena + scored->tagOffsets[t], (size_t)(scored->tagOffsets[t + 1] - scored->tagOffsets[t]));
                        S nn = e.NormalizeTag(raw);
                        if (!is_blank(nn)) list.push_back(Candidate{nn, scored->scores[t], order++, fixedSource >= 0 ? fixedSource : (int)scored->sources[t]});
                    }
                    std::stable_sort(list.begin(), list.end(), [](const Candidate& a, const Candidate& c) {
                        if (a.score != c.score) return a.score > c.score;
                        return a.order < c.order;
                    });
                    return list;
                };
                if (!scored->secondaryBegin) cands = parse(scored->tagBegin[i], scored->tagBegin[i + 1], -1);
                else {
                    // the two tagger calls and their merge, Program.cs:214-221
                    std::vector<Candidate> primary = parse(scored->tagBegin[i], scored->secondaryBegin[i], TP_SRC_PRIMARY_JOY);
                    std::vector<Candidate> secondary = parse(scored->secondaryBegin[i], scored->tagBegin[i + 1], TP_SRC_SECONDARY_JOY);
                    cands = Engine::MergeCandidateTags(primary, Engine::LimitSecondaryTags(primary, secondary, scored->secondaryCap));
                }
            }
            e.run_candidates(cands, rejected, caps[i], b.verdicts[i], wantDetail ? &b.details[i] : nullptr,
                             cx.firedWords ? &b.fired[i * cx.firedWords] : nullptr, cx.firedWords, nf ? &b.facetChoice[i * nf] : nullptr, &ctags[i]);
        }
    };
    if (threads <= 1 || n < 64) work(0, n);
    else {
        std::vector<std::thread> th;
        uint64_t per = (n + threads - 1) / threads;
        for (int t = 0; t < threads; ++t) { uint64_t lo = t * per, hi = std::min<uint64_t>(n, lo + per); if (lo < hi) th.emplace_back(work, lo, hi); }
        for (auto& x : th) x.join();
    }
    b.offsets.assign(n + 1, 0);
    for (uint64_t i = 0; i < n; ++i) b.offsets[i + 1] = b.offsets[i] + caps[i].size();
    b.captions.resize(b.offsets[n]);
    std::memset(&ctr, 0, sizeof ctr);
    for (uint64_t i = 0; i < n; ++i) {
        std::memcpy(b.captions.data() + b.offsets[i], caps[i].data(), caps[i].size());
        const tp_verdict& v = b.verdicts[i];
        ctr.files++;
        if (v.flags & TP_F_LOW) ctr.low++;
        if (v.flags & TP_F_ELIGIBLE) ctr.action_eligible++;
        if (v.flags & TP_F_FOOTJOB) ctr.action_footjob++;
        if (v.flags & TP_F_FALLBACK) ctr.fallback++;
        if (v.flags & TP_F_LINE_REJECTED) ctr.line_rejected++;
        if (v.flags & TP_F_SAFE_REJECT) ctr.safe_rejected++;
        for (auto& t : ctags[i]) b.hist[to8(t)]++;
    }
}

static std::string json_escape(const std::string& s) {
    std::string o;
    for (unsigned char c : s) {
        if (c == '"' || c == '\\') { o.push_back('\\'); o.push_back((char)c); }
        else if (c < 0x20) { char buf[8]; std::snprintf(buf, sizeof buf, "\\u%04x", c); o += buf; }
        else o.push_back((char)c);
    }
    return o;
}
static std::string json_list(const std::vector<S>& v) {
    std::string o = "[";
    for (size_t i = 0; i < v.size(); ++i) { if (i) o += ","; o += "\"" + json_escape(to8(v[i])) + "\""; }
    return o + "]";
}

}  // namespace orc

using namespace orc;

extern "C" {

struct orc_result {
    uint64_t n_files;
    const uint8_t* captions;
    const uint64_t* caption_offsets;
    const tp_verdict* verdicts;
    tp_counters counters;
    const uint32_t* fired;
    const uint16_t* facet_choice;
    void* opaque;
};

const char* orc_last_error(void) { return g_err.c_str(); }

void* orc_create(const tp_config* cfg) {
    try {
        if (!cfg) { g_err = "null config"; return nullptr; }
        std::unique_ptr<Ctx> cx(new Ctx());
        cx->eng.configure(*cfg);
        for (auto& r : cx->eng.ActiveActionRules) cx->nSubtypes += r.subs.size();
        cx->firedWords = (cx->eng.ActiveActionRules.size() + cx->nSubtypes + 31) / 32;
        return cx.release();
    } catch (const std::exception& ex) {
        g_err = std::string("Fatal error occurred. ") + ex.what();
        return nullptr;
    }
}
void orc_destroy(void* ctx) { delete (Ctx*)ctx; }

int orc_get_table_info(const void* ctx, tp_table_info* out) {
    const Ctx* cx = (const Ctx*)ctx;
    if (!cx || !out) return TP_E_INVALID_ARG;
    std::memset(out, 0, sizeof *out);
    out->n_action_rules = (uint32_t)cx->eng.ActiveActionRules.size();
    out->n_action_subtypes = (uint32_t)cx->nSubtypes;
    out->n_facets = (uint32_t)cx->eng.ActiveRequiredFacets.size();
    out->fired_words = (uint32_t)cx->firedWords;
    out->n_known_tags = (uint32_t)cx->eng.KnownTags.size();
    out->strict_known = cx->eng.ActiveStrictKnownTags;
    out->known_found = cx->eng.KnownTagDictionaryFound;
    return TP_OK;
}

static int finish(Batch* b, uint64_t n, const tp_counters& ctr, orc_result* out) {
    out->n_files = n;
    out->captions = b->captions.data();
    out->caption_offsets = b->offsets.data();
    out->verdicts = b->verdicts.data();
    out->counters = ctr;
    out->fired = b->fired.data();
    out->facet_choice = b->facetChoice.data();
    out->opaque = b;
    return TP_OK;
}

// threads <= 1: scalar; detail != 0 keeps the per-file lists for orc_detail_json.
int orc_run_lines(void* ctx, const uint8_t* arena, const uint64_t* offsets, uint64_t n, int threads, int detail, orc_result* out) {
    try {
        Ctx* cx = (Ctx*)ctx;
        if (!cx || !out || (n && (!arena || !offsets))) { g_err = "invalid argument"; return TP_E_INVALID_ARG; }
        std::unique_ptr<Batch> b(new Batch());
        tp_counters ctr;
        run_batch(*cx, arena, offsets, nullptr, n, detail != 0, threads, *b, ctr);
        return finish(b.release(), n, ctr, out);
    } catch (const std::exception& ex) { g_err = ex.what(); return TP_E_INVALID_ARG; }
}

int orc_run_scored(void* ctx, const uint8_t* arena, const uint64_t* tag_offsets, const uint64_t* tag_begin, const double* scores,
                   const uint8_t* sources, uint64_t n, int threads, int detail, orc_result* out) {
    try {
        Ctx* cx = (Ctx*)ctx;
        if (!cx || !out) { g_err = "invalid argument"; return TP_E_INVALID_ARG; }
        for (uint64_t t = 0, nt = n ? tag_begin[n] : 0; t < nt; ++t)
            if (!(scores[t] - scores[t] == 0.0)) { g_err = "scores must be finite"; return TP_E_INVALID_ARG; }   // NaN / infinity: not reachable through the reference's JSON ingest, left undefined
        std::unique_ptr<Batch> b(new Batch());
        tp_counters ctr;
        ScoredInput si{arena, tag_offsets, tag_begin, scores, sources, nullptr, 0};
        run_batch(*cx, nullptr, nullptr, &si, n, detail != 0, threads, *b, ctr);
        return finish(b.release(), n, ctr, out);
    } catch (const std::exception& ex) { g_err = ex.what(); return TP_E_INVALID_ARG; }
}

int orc_run_merged(void* ctx, const uint8_t* arena, const uint64_t* tag_offsets, const uint64_t* tag_begin, const uint64_t* secondary_begin,
                   const double* scores, int secondary_cap, uint64_t n, int threads, int detail, orc_result* out) {
    try {
        Ctx* cx = (Ctx*)ctx;
        if (!cx || !out || !secondary_begin) { g_err = "invalid argument"; return TP_E_INVALID_ARG; }
        for (uint64_t t = 0, nt = n ? tag_begin[n] : 0; t < nt; ++t)
            if (!(scores[t] - scores[t] == 0.0)) { g_err = "scores must be finite"; return TP_E_INVALID_ARG; }   // NaN / infinity: not reachable through the reference's JSON ingest, left undefined
        std::unique_ptr<Batch> b(new Batch());
        tp_counters ctr;
        ScoredInput si{arena, tag_offsets, tag_begin, scores, nullptr, secondary_begin, secondary_cap};
        run_batch(*cx, nullptr, nullptr, &si, n, detail != 0, threads, *b, ctr);
        return finish(b.release(), n, ctr, out);
    } catch (const std::exception& ex) { g_err = ex.what(); return TP_E_INVALID_ARG; }
}

// Cached tagger replies (JSON bodies), one or two per file: status receives 0 / 1 / 3 per file (see run_batch).
int orc_run_replies(void* ctx, const uint8_t* primary, const uint64_t* primary_offsets, const uint8_t* secondary, const uint64_t* secondary_offsets,
                    int secondary_cap, uint64_t n, int threads, int detail, uint8_t* status, orc_result* out) {
    try {
        Ctx* cx = (Ctx*)ctx;
        if (!cx || !out || !primary_offsets) { g_err = "invalid argument"; return TP_E_INVALID_ARG; }
        std::unique_ptr<Batch> b(new Batch());
        tp_counters ctr;
        ScoredInput si{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, secondary_cap};
        si.replies = primary; si.replyOffsets = primary_offsets; si.secondaryReplies = secondary; si.secondaryReplyOffsets = secondary_offsets; si.replyStatus = status;
        run_batch(*cx, nullptr, nullptr, &si, n, detail != 0, threads, *b, ctr);
        return finish(b.release(), n, ctr, out);
    } catch (const std::exception& ex) { g_err = ex.what(); return TP_E_INVALID_ARG; }
}

void orc_release(orc_result* r) {
    if (r && r->opaque) { delete (Batch*)r->opaque; r->opaque = nullptr; }
}

// "tag\tcount\n" for every caption tag of the batch, ordered by count descending then tag bytes ascending.
const char* orc_histogram_text(orc_result* r, uint64_t* len) {
    Batch* b = (Batch*)r->opaque;
    std::vector<std::pair<std::string, uint64_t>> v(b->hist.begin(), b->hist.end());
    std::sort(v.begin(), v.end(), [](const std::pair<std::string, uint64_t>& a, const std::pair<std::string, uint64_t>& c) {
        if (a.second != c.second) return a.second > c.second;
        return a.first < c.first;
    });
    b->histText.clear();
    for (auto& kv : v) { b->histText += kv.first; b->histText += '\t'; b->histText += std::to_string(kv.second); b->histText += '\n'; }
    if (len) *len = b->histText.size();
    return b->histText.c_str();
}

// One JSON object with the per-file lists the reference hands to AppendAudit (Program.cs:334); needs detail != 0.
const char* orc_detail_json(orc_result* r, uint64_t i) {
    Batch* b = (Batch*)r->opaque;
    if (i >= b->details.size()) return "{}";
    const FileDetail& d = b->details[i];
    b->detailText = "{\"final_tags\":" + json_list(d.finalTags) + ",\"caption_tags\":" + json_list(d.captionTags) + ",\"evidence\":" + json_list(d.evidence) +
                    ",\"fired\":" + json_list(d.fired) + ",\"dropped\":" + json_list(d.dropped) + ",\"dropped_male\":" + json_list(d.droppedMale) +
                    ",\"unknown\":" + json_list(d.unknown) + ",\"coverage\":" + json_list(d.covDetails) + ",\"stop_reason\":\"" + json_escape(to8(d.stopReason)) +
                    "\",\"reject_reason\":\"" + json_escape(to8(d.rejectReason)) + "\"}";
    return b->detailText.c_str();
}

// The line AppendAudit appends to __review/audit.csv for file i (Program.cs:966), without the newline; needs detail != 0.
// vocab_file null = KnownTagDictionaryPath null.
static void csv_field(const std::string& s, std::string& out) {   // Csv(), Program.cs:976-981
    out += '"';
    for (char c : s) { if (c == '"') out += '"'; out += c; }
    out += '"';
}
const char* orc_audit_row(orc_result* r, uint64_t i, const char* filename, const char* vocab_file, const char* used_profile) {
    Batch* b = (Batch*)r->opaque;
    if (i >= b->details.size()) return "";
    const FileDetail& d = b->details[i];
    auto semi = [](const std::vector<S>& v) { std::string o; for (size_t k = 0; k < v.size(); ++k) { if (k) o += ';'; o += to8(v[k]); } return o; };
    std::string& o = b->detailText;
    o.clear();
    csv_field(filename, o); o += ',';
    csv_field(semi(d.finalTags), o); o += ',';
    csv_field(semi(d.evidence), o); o += ',';
    csv_field(semi(d.fired), o); o += ',';
    csv_field(semi(d.dropped), o); o += ',';
    csv_field(to8(d.rejectReason), o); o += ',';          // rejectReason ?? ""
    csv_field(vocab_file ? vocab_file : "", o); o += ','; // vocabFile ?? ""
    csv_field(used_profile, o); o += ',';
    csv_field(to8(d.stopReason), o); o += ',';
    csv_field(semi(d.coverageTrace), o);
    return o.c_str();
}

// System.Text.Json's default encoder on a string (JsonSerializer.Serialize, Program.cs:874): Basic Latin stays except the
// control characters, DEL and " & ' + < > \ `; backslash and \b \f \n \r \t have short escapes; all else is \uXXXX (upper-case hex).
static void stj_string(const S& s, std::string& o) {
    static const char* const hex = "0123456789ABCDEF";
    o += '"';
    for (C c : s) {
        unsigned v = (unsigned)c;
        auto unit = [&]() { o += "\\u"; o += hex[(v >> 12) & 15]; o += hex[(v >> 8) & 15]; o += hex[(v >> 4) & 15]; o += hex[v & 15]; };
        if (v == '\\') o += "\\\\";
        else if (v == '\b') o += "\\b";
        else if (v == '\f') o += "\\f";
        else if (v == '\n') o += "\\n";
        else if (v == '\r') o += "\\r";
        else if (v == '\t') o += "\\t";
        else if (v < 0x20 || v >= 0x7F || v == '"' || v == '&' || v == '\'' || v == '+' || v == '<' || v == '>' || v == '`') unit();
        else o += (char)v;
    }
    o += '"';
}
static void stj_list(const std::vector<S>& v, std::string& o) {
    o += '[';
    for (size_t k = 0; k < v.size(); ++k) { if (k) o += ','; stj_string(v[k], o); }
    o += ']';
}
// The line AppendReviewLog appends for file i (Program.cs:871-890), without the newline: format 0 = review_candidates.jsonl,
// 1 = review_candidates.csv; "" when BuildReviewRecord returned null (Program.cs:336).  Needs detail != 0.
const char* orc_review_row(void* ctx, orc_result* r, uint64_t i, int format, const char* image_path, const char* caption_path, const char* thresholds_used,
                           const char* vocab_file) {
    Ctx* cx = (Ctx*)ctx;
    Batch* b = (Batch*)r->opaque;
    std::string& o = b->detailText;
    o.clear();
    if (i >= b->details.size() || !b->details[i].review.present) return o.c_str();
    const FileDetail& d = b->details[i];
    auto w = [](const char* p) { return utf8_to_16((const uint8_t*)p, std::strlen(p)); };
    auto semi = [](const std::vector<S>& v) { std::string x; for (size_t k = 0; k < v.size(); ++k) { if (k) x += ';'; x += to8(v[k]); } return x; };
    const bool found = cx->eng.KnownTagDictionaryFound;
    const std::string count = std::to_string(cx->eng.KnownTags.size());
    const std::string vocab = vocab_file ? vocab_file : "";                                  // KnownTagDictionaryPath ?? ""
    if (format == 0) {
        o += "{\"ImagePath\":"; stj_string(w(image_path), o);
        o += ",\"CaptionPath\":"; stj_string(w(caption_path), o);
        o += ",\"Reasons\":"; stj_list(d.review.reasons, o);
        o += ",\"BannedTokensDetected\":"; stj_list(d.review.banned, o);
        o += ",\"DroppedTags\":"; stj_list(d.review.dropped, o);
        o += ",\"FinalTagCount\":" + std::to_string(d.review.finalTagCount);
        o += std::string(",\"ShortCaption\":") + (d.review.shortCaption ? "true" : "false");
        o += ",\"ThresholdsUsed\":"; stj_string(w(thresholds_used), o);
        o += ",\"TopJoyTags\":"; stj_list(d.review.top, o);
        o += ",\"UnknownFinalTags\":"; stj_list(d.review.unknown, o);
        o += std::string(",\"KnownTagDictionaryFound\":") + (found ? "true" : "false");
        o += ",\"KnownTagDictionaryCount\":" + count;
        o += ",\"KnownTagDictionaryPath\":"; stj_string(w(vocab.c_str()), o);
        o += ",\"CoverageTrace\":"; stj_list(d.coverageTrace, o);
        o += ",\"CoverageStopReason\":"; stj_string(d.stopReason, o);
        o += '}';
    } else {
        csv_field(image_path, o); o += ','; csv_field(caption_path, o); o += ','; csv_field(semi(d.review.reasons), o); o += ',';
        csv_field(semi(d.review.banned), o); o += ','; csv_field(semi(d.review.dropped), o); o += ','; csv_field(std::to_string(d.review.finalTagCount), o); o += ',';
        csv_field(d.review.shortCaption ? "true" : "false", o); o += ','; csv_field(thresholds_used, o); o += ','; csv_field(semi(d.review.top), o); o += ',';
        csv_field(semi(d.review.unknown), o); o += ','; csv_field(found ? "true" : "false", o); o += ','; csv_field(count, o); o += ',';
        csv_field(vocab, o); o += ','; csv_field(to8(d.stopReason), o); o += ','; csv_field(semi(d.coverageTrace), o);
    }
    return o.c_str();
}

// Single-function probes used by the known-answer tests.
static thread_local std::string g_tmp;
const char* orc_normalize_tag(void* ctx, const uint8_t* s, uint64_t n, int basic) {
    Ctx* cx = (Ctx*)ctx;
    S in = utf8_to_16(s, (size_t)n);
    g_tmp = to8(basic ? Engine::NormalizeTagBasic(in) : cx->eng.NormalizeTag(in));
    return g_tmp.c_str();
}
int orc_classify_tag(void* ctx, const char* tag) { return (int)((Ctx*)ctx)->eng.ClassifyTag(utf8_to_16((const uint8_t*)tag, std::strlen(tag))); }
int orc_is_valid_tag(const char* tag) { return Engine::IsValidTag(utf8_to_16((const uint8_t*)tag, std::strlen(tag))) ? 1 : 0; }
int orc_should_drop_tag(void* ctx, const char* tag) { return ((Ctx*)ctx)->eng.ShouldDropTag(utf8_to_16((const uint8_t*)tag, std::strlen(tag))) ? 1 : 0; }
const char* orc_normalize_raw_tag_line(void* ctx, const uint8_t* s, uint64_t n) {
    g_tmp = to8(((Ctx*)ctx)->eng.NormalizeRawTagLine(utf8_to_16(s, (size_t)n)));
    return g_tmp.c_str();
}
// Existing-caption triage, Program.cs:205-207: bad[i] = LooksBad(NormalizeRawTagLine(text of file i)).
int orc_triage_lines(void* ctx, const uint8_t* arena, const uint64_t* offsets, uint64_t n, uint8_t* bad) {
    Ctx* cx = (Ctx*)ctx;
    for (uint64_t i = 0; i < n; ++i)
        bad[i] = Engine::LooksBad(cx->eng.NormalizeRawTagLine(utf8_to_16(arena + offsets[i], (size_t)(offsets[i + 1] - offsets[i])))) ? 1 : 0;
    return 0;
}
// InferActionTags on explicit lists (self-check items 5-8): tags separated by '\n'; returns the inferred list, '\n'-joined.
const char* orc_infer_action_tags(void* ctx, const char* final_tags, const char* evidence_tags) {
    Ctx* cx = (Ctx*)ctx;
    auto split = [](const char* p) { std::vector<S> v; std::string cur; for (;; ++p) { if (*p == '\n' || !*p) { if (!cur.empty()) v.push_back(utf8_to_16((const uint8_t*)cur.data(), cur.size())); cur.clear(); if (!*p) break; } else cur.push_back(*p); } return v; };
    std::vector<S> ft = split(final_tags);
    ISet ev;
    for (auto& t : split(evidence_tags)) ev.add(t);
    std::vector<S> fired;
    std::vector<S> inf = cx->eng.InferActionTags(ft, ev, fired, nullptr);
    g_tmp.clear();
    for (size_t i = 0; i < inf.size(); ++i) { if (i) g_tmp += '\n'; g_tmp += to8(inf[i]); }
    return g_tmp.c_str();
}
const char* orc_format_caption(void* ctx, const char* final_tags) {
    Ctx* cx = (Ctx*)ctx;
    std::vector<S> v; std::string cur;
    for (const char* p = final_tags;; ++p) { if (*p == '\n' || !*p) { if (!cur.empty()) v.push_back(utf8_to_16((const uint8_t*)cur.data(), cur.size())); cur.clear(); if (!*p) break; } else cur.push_back(*p); }
    std::vector<S> c = cx->eng.CaptionTags(v);
    g_tmp.clear();
    for (size_t i = 0; i < c.size(); ++i) { if (i) g_tmp += ", "; g_tmp += to8(c[i]); }
    return g_tmp.c_str();
}
// SelectTags(ParseTagCandidatesFromLine(line, null, weight, OpenAI)).Tags, ", "-joined (self-check items 1-4, Program.cs:1958-1959).
const char* orc_select_tags_from_line(void* ctx, const uint8_t* s, uint64_t n, double weight) {
    Ctx* cx = (Ctx*)ctx;
    Selection sel = cx->eng.SelectTags(cx->eng.ParseTagCandidatesFromLine(utf8_to_16(s, (size_t)n), nullptr, weight, TP_SRC_OPENAI));
    g_tmp.clear();
    for (size_t i = 0; i < sel.tags.size(); ++i) { if (i) g_tmp += ", "; g_tmp += to8(sel.tags[i]); }
    return g_tmp.c_str();
}
const char* orc_minimal_fallback(void* ctx) { g_tmp = to8(((Ctx*)ctx)->eng.minimalFallback); return g_tmp.c_str(); }
const char* orc_trigger(void* ctx) { g_tmp = to8(((Ctx*)ctx)->eng.Trigger); return g_tmp.c_str(); }

// Names behind the fired bitmap / facet_choice arrays, as a JSON object.
const char* orc_rule_names_json(void* ctx) {
    Ctx* cx = (Ctx*)ctx;
    std::string o = "{\"actions\":[";
    bool first = true;
    for (auto& r : cx->eng.ActiveActionRules) { if (!first) o += ","; first = false; o += "\"" + json_escape(to8(r.parent)) + "\""; }
    o += "],\"subtypes\":[";
    first = true;
    for (auto& r : cx->eng.ActiveActionRules) for (auto& s : r.subs) { if (!first) o += ","; first = false; o += "\"" + json_escape(to8(r.parent + u(":") + s.tag)) + "\""; }
    o += "],\"facets\":[";
    first = true;
    for (auto& f : cx->eng.ActiveRequiredFacets) {
        if (!first) o += ",";
        first = false;
        std::vector<S> rt; for (auto& r : f.rules) rt.push_back(r.tag);
        o += "{\"name\":\"" + json_escape(to8(f.name)) + "\",\"default\":\"" + json_escape(to8(f.def)) + "\",\"rules\":" + json_list(rt) + "}";
    }
    o += "]}";
    g_tmp = o;
    return g_tmp.c_str();
}

}  // extern "C"

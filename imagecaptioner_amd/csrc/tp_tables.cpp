// Host-side table compiler (see tp_tables.hpp).
//
// Loader semantics follow Program.cs line by line, including the load-order
// quirk: allow/deny lists, the vocabulary and the profile are normalised with
// the BUILT-IN default lexicon (Program.cs:159-161 run before :163), while the
// per-string attributes the kernels read are computed with the final one.
#include "tp_tables.hpp"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iterator>
#include <map>
#include <unordered_map>
#include <unordered_set>

#include "tp_json.hpp"

namespace tp {
namespace {

typedef std::u16string W;
typedef char16_t wc;

#include "tp_case_tables.inc"

wc case_map(const uint32_t* tab, size_t n, wc c) {
    size_t lo = 0, hi = n;
    while (lo < hi) {
        size_t mid = (lo + hi) / 2;
        uint32_t key = tab[mid] >> 16;
        if (key == c) return (wc)(tab[mid] & 0xFFFF);
        if (key < c) lo = mid + 1; else hi = mid;
    }
    return c;
}
wc lower_invariant(wc c) {   // char.ToLowerInvariant
    if (c < 0x80) return (c >= 'A' && c <= 'Z') ? (wc)(c | 0x20) : c;
    return case_map(kCaseLower, sizeof kCaseLower / sizeof kCaseLower[0], c);
}
wc upper_ordinal(wc c) {     // the folding StringComparer.OrdinalIgnoreCase compares with
    if (c < 0x80) return (c >= 'a' && c <= 'z') ? (wc)(c & ~0x20) : c;
    return case_map(kCaseUpper, sizeof kCaseUpper / sizeof kCaseUpper[0], c);
}
bool is_space(wc c) {        // char.IsWhiteSpace (also what \s matches)
    switch (c) {
        case 0x20: case 0x85: case 0xA0: case 0x1680: case 0x2028: case 0x2029: case 0x202F: case 0x205F: case 0x3000: return true;
        default: return (c >= 0x09 && c <= 0x0D) || (c >= 0x2000 && c <= 0x200A);
    }
}
W lowered(W s) { for (auto& c : s) c = lower_invariant(c); return s; }
W folded(W s) { for (auto& c : s) c = upper_ordinal(c); return s; }
bool blank(const W& s) { return std::all_of(s.begin(), s.end(), is_space); }
W trimmed(const W& s) {
    size_t a = 0, b = s.size();
    while (a < b && is_space(s[a])) ++a;
    while (b > a && is_space(s[b - 1])) --b;
    return s.substr(a, b - a);
}
W widen(const char* ascii) { W w; while (*ascii) w.push_back((wc)(unsigned char)*ascii++); return w; }

W from_utf8(const uint8_t* p, size_t n) {
    W out;
    for (size_t i = 0; i < n;) {
        uint32_t b = p[i];
        if (b < 0x80) { out.push_back((wc)b); ++i; continue; }
        int extra = (b >= 0xF0 && b <= 0xF4) ? 3 : (b >= 0xE0) ? 2 : (b >= 0xC2) ? 1 : -1;
        uint32_t cp = 0;
        bool good = extra > 0 && i + extra < n;
        if (good) {
            cp = extra == 1 ? (b & 0x1F) : extra == 2 ? (b & 0x0F) : (b & 0x07);
            for (int k = 1; k <= extra && good; ++k) {
                if ((p[i + k] & 0xC0) != 0x80) good = false; else cp = (cp << 6) | (p[i + k] & 0x3F);
            }
            if (good && ((extra == 2 && cp < 0x800) || (extra == 3 && (cp < 0x10000 || cp > 0x10FFFF)) || (cp >= 0xD800 && cp <= 0xDFFF))) good = false;
        }
        if (!good) { out.push_back(0xFFFD); ++i; continue; }
        if (cp >= 0x10000) { cp -= 0x10000; out.push_back((wc)(0xD800 | (cp >> 10))); out.push_back((wc)(0xDC00 | (cp & 0x3FF))); }
        else out.push_back((wc)cp);
        i += extra + 1;
    }
    return out;
}
std::string to_utf8(const W& s) {
    std::string o;
    for (size_t i = 0; i < s.size(); ++i) {
        uint32_t cp = s[i];
        if (cp >= 0xD800 && cp < 0xDC00 && i + 1 < s.size() && s[i + 1] >= 0xDC00 && s[i + 1] < 0xE000) { cp = 0x10000 + ((cp - 0xD800) << 10) + (s[i + 1] - 0xDC00); ++i; }
        else if (cp >= 0xD800 && cp < 0xE000) cp = 0xFFFD;
        if (cp < 0x80) o += (char)cp;
        else if (cp < 0x800) { o += (char)(0xC0 | (cp >> 6)); o += (char)(0x80 | (cp & 0x3F)); }
        else if (cp < 0x10000) { o += (char)(0xE0 | (cp >> 12)); o += (char)(0x80 | ((cp >> 6) & 0x3F)); o += (char)(0x80 | (cp & 0x3F)); }
        else { o += (char)(0xF0 | (cp >> 18)); o += (char)(0x80 | ((cp >> 12) & 0x3F)); o += (char)(0x80 | ((cp >> 6) & 0x3F)); o += (char)(0x80 | (cp & 0x3F)); }
    }
    return o;
}

// Case-insensitive containers keyed by the ordinal folding.
struct FoldSet {
    std::unordered_set<W> keys;
    std::vector<W> order;
    bool insert(const W& s) { if (!keys.insert(folded(s)).second) return false; order.push_back(s); return true; }
    bool contains(const W& s) const { return keys.count(folded(s)) != 0; }
    size_t size() const { return order.size(); }
};
template <class V>
struct FoldMap {
    std::unordered_map<W, std::pair<W, V>> m;   // folded key -> (key as first inserted, value); `d[k] = v` keeps the first key
    void put(const W& k, const V& v) {
        auto it = m.find(folded(k));
        if (it == m.end()) m.emplace(folded(k), std::make_pair(k, v)); else it->second.second = v;
    }
    const V* get(const W& k) const { auto it = m.find(folded(k)); return it == m.end() ? nullptr : &it->second.second; }
};
std::vector<W> unique_fold(const std::vector<W>& v) { FoldSet s; for (auto& x : v) s.insert(x); return s.order; }

// ---- NormalizeTagBasic, Program.cs:1267-1273 --------------------------------------------------------
W normalize_basic(const W& raw) {
    W t = trimmed(raw);
    size_t a = 0, b = t.size();
    auto quote = [](wc c) { return c == '"' || c == '\'' || c == '`'; };
    while (a < b && quote(t[a])) ++a;
    while (b > a && quote(t[b - 1])) --b;
    auto punct = [](wc c) { return c == ';' || c == '.' || c == 0x3002 || c == 0x3001; };
    if (b > a && t[b - 1] == '\n') {          // `$` also matches before a final '\n': that run goes, the '\n' stays
        size_t q = b - 1;
        while (q > a && punct(t[q - 1])) --q;
        t = t.substr(a, q - a) + W(1, (wc)'\n');
    } else {
        while (b > a && punct(t[b - 1])) --b;
        t = t.substr(a, b - a);
    }
    W out;
    for (wc c : t) {
        wc e = is_space(c) ? (wc)'_' : c;
        if (e == '_' && !out.empty() && out.back() == '_') continue;
        out.push_back(e);
    }
    return lowered(out);
}

struct ActionSub { W tag; std::vector<W> any, all; };
struct Action { W parent; std::vector<W> all, any, forb; std::vector<ActionSub> subs; };
struct FacetR { W tag; std::vector<W> any, all; };
struct FacetDef { W name, def; std::vector<FacetR> rules; };

bool read_source(const tp_source& src, std::vector<uint8_t>& out) {
    out.clear();
    if (src.data) { out.assign(src.data, src.data + src.len); return true; }
    if (!src.path || !src.path[0]) return false;
    std::ifstream f(src.path, std::ios::binary);
    if (!f.good()) return false;
    out.assign(std::istreambuf_iterator<char>(f), std::istreambuf_iterator<char>());
    return true;
}
W text_of(const std::vector<uint8_t>& b) {
    size_t skip = (b.size() >= 3 && b[0] == 0xEF && b[1] == 0xBB && b[2] == 0xBF) ? 3 : 0;
    return from_utf8(b.data() + skip, b.size() - skip);
}
std::vector<W> lines_of(const W& text) {     // File.ReadLines: "\n", "\r" and "\r\n" end a line
    std::vector<W> out;
    W cur;
    bool open = false;
    for (size_t i = 0; i < text.size(); ++i) {
        wc c = text[i];
        if (c == '\n' || c == '\r') {
            if (c == '\r' && i + 1 < text.size() && text[i + 1] == '\n') ++i;
            out.push_back(cur);
            cur.clear();
            open = false;
        } else { cur.push_back(c); open = true; }
    }
    if (open) out.push_back(cur);
    return out;
}
std::vector<W> split_tags(const W& s) {       // SplitTags, Program.cs:1181-1187
    std::vector<W> out;
    if (blank(s)) return out;
    size_t start = 0;
    while (true) {
        size_t comma = s.find((wc)',', start);
        W part = trimmed(s.substr(start, comma == W::npos ? W::npos : comma - start));
        if (!part.empty() && !blank(part)) out.push_back(part);
        if (comma == W::npos) break;
        start = comma + 1;
    }
    return out;
}

const char* const kSafeList[20] = {"nsfw", "nude", "naked", "nipples", "breasts", "pussy", "cameltoe", "sex", "penis", "cum", "orgasm", "masturbation",
                                   "fellatio", "anal", "vaginal", "hentai", "explicit", "porn", "bdsm", "futanari"};   // Program.cs:43-45

struct ClassLits { uint32_t bit; const char* alts; };
const ClassLits kClassLits[] = {
    {CM_FACE, "hair|eyes|eyebrows|eyelashes|pupil|bangs|twintails|ponytail|braid|ahoge|blush|smile|frown|angry|surprised|open_mouth|closed_mouth|tears|looking_"},   // Program.cs:48
    {CM_OUTFIT, "dress|skirt|shirt|jacket|coat|hoodie|sweater|uniform|kimono|swimsuit|socks|stockings|pantyhose|shoes|boots|gloves|hat|ribbon|bow|tie|neckwear|blazer|pants|shorts"},   // :49
    {CM_POSE, "pose|standing|sitting|lying|kneeling|crouching|walking|running|jumping|leaning|holding|raising|arms|hands|fingers|looking_at_viewer|looking_away|from_side|from_behind"},   // :50
    {CM_BACKGROUND, "indoors|outdoors|room|bedroom|street|city|school|classroom|park|forest|beach|sky|night|sunset|window|bed|chair|office|kitchen|bathroom"},   // :51
    {CM_STYLE, "lighting|backlighting|rim_light|soft_light|hard_light|shadow|bloom|depth_of_field|bokeh|monochrome|highres|absurdres|masterpiece|best_quality|high_quality|detailed"},   // :52
    {CM_HAIR, "hair|bangs|twintails|ponytail|braid|ahoge"},   // :53
    {CM_EYES, "eyes|eyelashes|eyebrows|pupil|heterochromia"},   // :54
    {CM_SORRY, "sorry"},   // :57
    {CM_WATERMARK, "watermark"},   // :1130
};

// Aho-Corasick automaton over the class literals, as a dense goto table.
void build_dfa(std::vector<uint16_t>& next, std::vector<uint16_t>& outm) {
    std::vector<std::vector<int>> go(1, std::vector<int>(DFA_SYMS, -1));
    std::vector<uint32_t> out(1, 0);
    for (const ClassLits& cl : kClassLits) {
        const char* p = cl.alts;
        int st = 0;
        for (;; ++p) {
            if (*p == '|' || *p == 0) {
                out[st] |= cl.bit;
                st = 0;
                if (!*p) break;
                continue;
            }
            uint32_t sy = dfa_sym((uint8_t)*p);
            if (go[st][sy] < 0) { go[st][sy] = (int)go.size(); go.emplace_back(DFA_SYMS, -1); out.push_back(0); }
            st = go[st][sy];
        }
    }
    size_t ns = go.size();
    if (ns > 65535) throw ConfigError("classifier automaton too large");
    std::vector<int> fail(ns, 0), queue;
    for (int sy = 0; sy < DFA_SYMS; ++sy) {
        if (go[0][sy] < 0) go[0][sy] = 0; else { fail[go[0][sy]] = 0; queue.push_back(go[0][sy]); }
    }
    for (size_t qi = 0; qi < queue.size(); ++qi) {
        int s = queue[qi];
        out[s] |= out[fail[s]];
        for (int sy = 0; sy < DFA_SYMS; ++sy) {
            int nx = go[s][sy];
            if (nx < 0) go[s][sy] = go[fail[s]][sy];
            else { fail[nx] = go[fail[s]][sy]; queue.push_back(nx); }
        }
    }
    next.resize(ns * DFA_SYMS);
    outm.resize(ns);
    for (size_t s = 0; s < ns; ++s) {
        outm[s] = (uint16_t)out[s];
        for (int sy = 0; sy < DFA_SYMS; ++sy) next[s * DFA_SYMS + sy] = (uint16_t)go[s][sy];
    }
}
uint32_t class_mask(const std::vector<uint16_t>& next, const std::vector<uint16_t>& outm, const W& lower_text) {
    uint32_t st = 0, mask = 0;
    for (wc c : lower_text) {
        uint32_t sy = c < 0x80 ? dfa_sym((uint8_t)c) : 27u;
        st = next[st * DFA_SYMS + sy];
        mask |= outm[st];
    }
    return mask;
}

// The loaded state of Program (its static fields), built in Main's order.
struct Loader {
    std::vector<uint16_t> dfa_next, dfa_out;
    FoldSet subject, banned, builtin_deny, action_tags, safe_reject, allow, deny, known, force_allow;
    FoldMap<W> builtin_aliases;              // ActionTagAliases, Program.cs:71-78
    FoldMap<W> aliases;                      // ActiveLexicon.Aliases
    FoldMap<std::vector<W>> implied;         // ActiveLexicon.Implied
    std::vector<Action> actions;
    std::vector<FacetDef> facets;
    bool known_found = false, strict = false, strict_explicit = false;
    bool has_profile = false;
    std::vector<W> always_add;
    W trigger;

    Loader() {
        build_dfa(dfa_next, dfa_out);
        for (auto s : {"1girl", "1boy", "solo", "multiple_girls", "multiple_boys", "2girls", "2boys", "girl", "boy"}) subject.insert(widen(s));       // :27-29
        for (auto s : {"1boy", "boy", "multiple_boys", "2boys", "3boys", "4boys", "male_focus", "yaoi", "male"}) banned.insert(widen(s));          // :33-35
        for (auto s : {"watermark", "signature", "text", "logo", "jpeg_artifacts", "lowres", "worst_quality", "rating", "artist_name"}) builtin_deny.insert(widen(s));   // :60-62
        for (auto s : {"footjob", "penis_between_feet", "foot_on_penis", "foot_sole_stimulation", "trampling_penis", "handjob", "penis_in_hand", "hand_wrapping_penis",
                       "squeezing_penis", "sex", "vaginal_penetration", "missionary", "doggystyle", "cowgirl_position", "blowjob", "penis_in_mouth", "fellatio"})
            action_tags.insert(widen(s));                                                                                                              // :64-69
        for (auto s : kSafeList) safe_reject.insert(widen(s));
        builtin_aliases.put(widen("pedal_stimulation"), widen("footjob"));
        builtin_aliases.put(widen("feet_stimulation"), widen("footjob"));
        builtin_aliases.put(widen("foot_stimulation"), widen("footjob"));
        builtin_aliases.put(widen("sexual_intercourse"), widen("sex"));
        builtin_aliases.put(widen("paizuri_sex"), widen("sex"));
        builtin_aliases.put(widen("fellatio"), widen("blowjob"));
        default_lexicon();
        actions = builtin_actions();
    }

    // BuildDefaultLexicon, Program.cs:1228-1253
    void default_lexicon() {
        aliases = FoldMap<W>();
        implied = FoldMap<std::vector<W>>();
        for (auto& kv : builtin_aliases.m) aliases.put(normalize_basic(kv.second.first), normalize_basic(kv.second.second));
        const char* al[][2] = {{"sole", "foot_sole"}, {"soles", "foot_sole"}, {"foot", "feet"}, {"hands", "hand"}, {"fingers", "hand"}, {"licking", "oral"}};
        for (auto& p : al) aliases.put(widen(p[0]), widen(p[1]));
        auto imp = [&](const char* k, std::initializer_list<const char*> v) { std::vector<W> l; for (auto s : v) l.push_back(widen(s)); implied.put(widen(k), l); };
        imp("barefoot", {"feet"}); imp("foot", {"feet"}); imp("toes", {"feet"}); imp("sole", {"feet"}); imp("soles", {"feet"}); imp("foot_sole", {"feet"});
        imp("hands", {"hand"}); imp("fingers", {"hand"}); imp("oral", {"mouth"}); imp("licking", {"oral", "mouth"});
    }
    // ActionInferenceRules, Program.cs:80-126
    static std::vector<Action> builtin_actions() {
        auto L = [](std::initializer_list<const char*> v) { std::vector<W> l; for (auto s : v) l.push_back(widen(s)); return l; };
        auto S = [&](const char* tag, std::initializer_list<const char*> any) { return ActionSub{widen(tag), L(any), {}}; };
        return {
            Action{widen("footjob"), L({"feet", "penis"}), {}, {},
                   {S("penis_between_feet", {"between_feet", "between", "sandwiched", "foot_sandwich"}), S("foot_on_penis", {"foot_on_penis", "footjob", "foot_press", "foot_pressing"}),
                    S("foot_sole_stimulation", {"sole", "foot_sole", "sole_focus", "underside"}), S("trampling_penis", {"stepping", "trampling", "step_on", "weight_on_penis"})}},
            Action{widen("handjob"), L({"penis", "hand"}), {}, L({"vaginal_penetration", "anal_penetration", "vaginal", "anal", "insertion", "penetration"}),
                   {S("penis_in_hand", {"penis_in_hand", "holding_penis", "holding"}), S("hand_wrapping_penis", {"grip", "gripping", "hand_wrapping_penis", "wrapped_hand"}),
                    S("squeezing_penis", {"squeezing", "stroking", "jerking", "pumping_motion"})}},
            Action{widen("sex"), L({"penis"}), L({"vagina", "vaginal", "vaginal_penetration", "insertion", "penetration", "intercourse"}), {},
                   {S("vaginal_penetration", {"vaginal_penetration", "vaginal", "penis_in_vagina", "insertion", "penetration"}), S("missionary", {"missionary", "missionary_position", "face_to_face"}),
                    S("doggystyle", {"doggystyle", "doggy_style", "from_behind", "rear_entry"}), S("cowgirl_position", {"cowgirl", "cowgirl_position", "female_on_top"})}},
            Action{widen("blowjob"), L({"penis"}), L({"mouth", "oral", "fellatio", "sucking"}), {},
                   {S("penis_in_mouth", {"penis_in_mouth", "deepthroat", "mouth_penetration"}), S("fellatio", {"fellatio", "oral", "oral_sex"})}},
        };
    }
    // BuildDefaultRequiredFacets, Program.cs:1255-1265
    static std::vector<FacetDef> builtin_facets() {
        auto L = [](std::initializer_list<const char*> v) { std::vector<W> l; for (auto s : v) l.push_back(widen(s)); return l; };
        auto R = [&](const char* tag, std::initializer_list<const char*> any) { return FacetR{widen(tag), L(any), {}}; };
        return {
            FacetDef{widen("viewpoint"), widen("unknown_pov"), {R("male_pov", {"male_pov", "pov_male", "from_male_pov"}), R("third_person", {"third_person", "third_person_view"})}},
            FacetDef{widen("gaze"), widen("unknown_gaze"), {R("looking_at_camera", {"looking_at_viewer", "looking_at_camera"}), R("looking_at_partner", {"looking_at_male", "looking_at_partner"}), R("looking_at_object", {"looking_at_penis", "looking_at_object"})}},
            FacetDef{widen("underwear"), widen("unknown_underwear"), {R("panties_aside", {"panties_aside"}), R("panties_off", {"panties_off", "no_panties"}), R("panties_on", {"panties_on", "wearing_panties"})}},
            FacetDef{widen("fluids"), widen("unknown_fluids"), {R("semen_present", {"semen", "cum"}), R("precum_present", {"pre_ejaculate", "precum"}), R("no_fluids", {"no_fluids"})}},
            FacetDef{widen("ejaculation"), widen("unknown_ejaculation"), {R("ejaculation", {"ejaculation", "cumshot"}), R("no_ejaculation", {"no_ejaculation"})}},
            FacetDef{widen("expression"), widen("unknown_expression"), {R("ahegao", {"ahegao"}), R("smile", {"smile", "smiling"}), R("neutral", {"neutral_expression"})}},
            FacetDef{widen("mouth_shape"), widen("unknown_mouth"), {R("tongue_out", {"tongue_out"}), R("mouth_open", {"open_mouth"}), R("mouth_closed", {"closed_mouth"})}},
        };
    }

    // the alias lookups of NormalizeTag, Program.cs:1155-1156
    W alias_of(const W& t) const {
        if (const W* v = aliases.get(t)) return *v;
        if (const W* v = builtin_aliases.get(t)) return *v;
        return t;
    }
    W normalize(const W& raw) const { return alias_of(normalize_basic(raw)); }   // NormalizeTag, Program.cs:1149-1158

    static bool valid_tag(const W& s) {                                           // IsValidTag, Program.cs:1135-1139
        if (blank(s) || s[0] == ':') return false;
        size_t n = s.size();
        if (s[n - 1] == '\n') --n;                                                // `$` accepts a final '\n'
        if (n == 0) return false;
        for (size_t i = 0; i < n; ++i) {
            wc c = s[i];
            if (c >= 0x80) return false;
            if (!(i == 0 ? is_tag_first8((uint8_t)c) : is_tag_rest8((uint8_t)c))) return false;
        }
        return true;
    }
    static bool has(const W& s, const char* needle) { return s.find(widen(needle)) != W::npos; }
    static bool has_looks_bad_phrase(const W& lower) {                            // DropTagsContains.Any(lower.Contains), Program.cs:56-58
        for (auto s : {"negative:", "positive:", "sorry", "i can't", "i cannot", "cannot provide", "unable to", "as an ai", "i'm not able", "i won't"})
            if (has(lower, s)) return true;
        W w = widen("i won"); w.push_back(0x2019); w.push_back('t');
        return lower.find(w) != W::npos;
    }
    bool should_drop(const W& t) const {                                          // ShouldDropTag, Program.cs:1121-1133
        W lower = lowered(t);
        if (force_allow.contains(lower)) return false;
        if (!valid_tag(lower)) return true;
        if (has_looks_bad_phrase(lower)) return true;
        if (lower.compare(0, 7, widen("rating:")) == 0) return true;
        if (builtin_deny.contains(lower) || banned.contains(lower) || deny.contains(lower)) return true;
        if (allow.size() > 0 && !allow.contains(lower) && (lower.compare(0, 6, widen("rating")) == 0 || has(lower, "watermark"))) return true;
        if (known_found && strict && !known.contains(lower)) return true;
        return false;
    }
    bool name_like(const W& s) const {                                            // CharacterNameRegex, Program.cs:47
        size_t n = s.size();
        if (n && s[n - 1] == '\n') --n;
        int segs = 0;
        size_t run = 0;
        for (size_t i = 0; i < n; ++i) {
            wc c = s[i];
            bool alnum = (c >= 'a' && c <= 'z') || (c >= 'A' && c <= 'Z') || (c >= '0' && c <= '9');
            if (alnum) { if (run == 0) ++segs; ++run; }
            else if (c == '_') { if (run == 0) return false; run = 0; }
            else return false;
        }
        return run > 0 && segs >= 2 && segs <= 4;
    }
    uint32_t bucket_of(const W& s, uint32_t mask) const {                         // ClassifyTag + IsLikelyCharacterName, Program.cs:1103-1119
        bool any5 = mask & (CM_FACE | CM_OUTFIT | CM_POSE | CM_BACKGROUND | CM_STYLE);
        if (subject.contains(s) || (name_like(s) && !any5)) return B_IDENTITY;
        if (mask & CM_FACE) return B_FACE;
        if (mask & CM_OUTFIT) return B_OUTFIT;
        if ((mask & CM_POSE) || action_tags.contains(s)) return B_POSE;
        if (mask & CM_BACKGROUND) return B_BACKGROUND;
        if (mask & CM_STYLE) return B_STYLE;
        return B_OTHER;
    }

    // LoadTagFile, Program.cs:1429-1439
    FoldSet load_tag_file(const tp_source& src) const {
        FoldSet set;
        std::vector<uint8_t> bytes;
        if (!read_source(src, bytes)) return set;
        for (auto& line : lines_of(text_of(bytes))) {
            W t = trimmed(line);
            if (blank(t) || t[0] == '#') continue;
            set.insert(normalize(t));
        }
        return set;
    }
    // ExtractKnownTagsFromJson, Program.cs:1036-1053
    void known_from_json(const JDoc& d, int32_t n, FoldSet& tags) const {
        switch (d.kind(n)) {
            case JKind::String: { W t = normalize(d.str(n)); if (!blank(t)) tags.insert(t); break; }
            case JKind::Array: for (int32_t c = d.nodes[n].first; c >= 0; c = d.nodes[c].next) known_from_json(d, c, tags); break;
            case JKind::Object:
                for (int32_t c = d.nodes[n].first; c >= 0; c = d.nodes[c].next) {
                    known_from_json(d, c, tags);
                    W t = normalize(d.member_name(c));
                    if (!blank(t)) tags.insert(t);
                }
                break;
            default: break;
        }
    }
    // LoadKnownTagDictionary + TryLoadKnownTagsFromFile, Program.cs:983-1034
    void load_vocabulary(const tp_config& cfg) {
        known = FoldSet();
        known_found = false;
        std::vector<uint8_t> bytes;
        if (read_source(cfg.vocabulary, bytes)) {
            bool json = cfg.vocabulary_is_json == 1;
            if (cfg.vocabulary_is_json < 0 && cfg.vocabulary.path) {
                std::string p = cfg.vocabulary.path;
                size_t dot = p.rfind('.');
                std::string ext = dot == std::string::npos ? std::string() : p.substr(dot);
                std::transform(ext.begin(), ext.end(), ext.begin(), [](unsigned char ch) { return (char)std::tolower(ch); });
                json = ext == ".json";
            }
            FoldSet loaded;
            bool ok = true;
            try {
                if (json) {
                    JDoc d = JDoc::parse(bytes.data(), bytes.size());
                    known_from_json(d, d.root, loaded);
                } else {
                    for (auto& line : lines_of(text_of(bytes))) {
                        W cleaned = trimmed(line);
                        if (blank(cleaned)) continue;
                        size_t start = 0;
                        while (true) {
                            size_t cut = cleaned.find_first_of(u",;", start);
                            W piece = trimmed(cleaned.substr(start, cut == W::npos ? W::npos : cut - start));
                            if (!piece.empty()) { W t = normalize(piece); if (!blank(t)) loaded.insert(t); }
                            if (cut == W::npos) break;
                            start = cut + 1;
                        }
                    }
                }
            } catch (const JsonFail&) { ok = false; }
            if (ok && loaded.size() > 0) {
                known = loaded;
                known_found = true;
                if (!strict_explicit) strict = true;
                return;
            }
        }
        if (!strict_explicit) strict = false;
    }
    // LoadProfile, Program.cs:918-931.  Only keys equal to Trigger / AlwaysAdd / PreferKeep ignoring case bind.
    void load_profile(const tp_config& cfg) {
        has_profile = false;
        always_add.clear();
        std::vector<uint8_t> bytes;
        if (!read_source(cfg.profile, bytes)) return;
        W trig = widen("sgrui");
        bool trig_null = false, list_broken = false;
        std::vector<W> always;
        try {
            JDoc d = JDoc::parse(bytes.data(), bytes.size());
            if (d.kind(d.root) != JKind::Object) return;       // null -> no profile; any other kind fails to bind
            auto is_key = [&](int32_t c, const char* want) { return folded(d.member_name(c)) == folded(widen(want)); };
            auto bind_list = [&](int32_t c, std::vector<W>* into, bool* broken) {
                if (into) into->clear();
                if (broken) *broken = false;
                if (d.kind(c) == JKind::Null) { if (broken) *broken = true; return; }
                if (d.kind(c) != JKind::Array) throw JsonFail("list expected");
                for (int32_t e = d.nodes[c].first; e >= 0; e = d.nodes[e].next) {
                    if (d.kind(e) == JKind::Null) { if (broken) *broken = true; }
                    else if (d.kind(e) == JKind::String) { if (into) into->push_back(d.str(e)); }
                    else throw JsonFail("string expected");
                }
            };
            for (int32_t c = d.nodes[d.root].first; c >= 0; c = d.nodes[c].next) {
                if (is_key(c, "Trigger")) {
                    if (d.kind(c) == JKind::Null) trig_null = true;
                    else if (d.kind(c) == JKind::String) { trig = d.str(c); trig_null = false; }
                    else throw JsonFail("string expected");
                } else if (is_key(c, "AlwaysAdd")) bind_list(c, &always, &list_broken);
                else if (is_key(c, "PreferKeep")) bind_list(c, nullptr, nullptr);
            }
        } catch (const JsonFail&) { return; }
        if (!trig_null && !blank(trig)) trigger = normalize(trig);                 // Program.cs:925
        if (list_broken) return;                                                  // Program.cs:926 throws on a null list or element: caught, no profile
        std::vector<W> aa;
        for (auto& t : always) { W n = normalize(t); if (!blank(n)) aa.push_back(n); }
        always_add = unique_fold(aa);
        has_profile = true;
    }

    // ParseStringArray, Program.cs:1408-1415
    static std::vector<W> string_array(const JDoc& d, int32_t n) {
        std::vector<W> out;
        if (d.kind(n) == JKind::String) { W s = normalize_basic(d.str(n)); if (!blank(s)) out.push_back(s); return out; }
        if (d.kind(n) != JKind::Array) return out;
        for (int32_t c = d.nodes[n].first; c >= 0; c = d.nodes[c].next) { W s = normalize_basic(d.string_or_empty(c)); if (!blank(s)) out.push_back(s); }
        return unique_fold(out);
    }
    static std::vector<W> opt_array(const JDoc& d, int32_t obj, const char16_t* key) {
        int32_t m = d.member(obj, key);
        return m < 0 ? std::vector<W>() : string_array(d, m);
    }
    // LoadLexicon, Program.cs:1275-1305 (merges over the defaults)
    void load_lexicon(const tp_source& src) {
        std::vector<uint8_t> bytes;
        if (!read_source(src, bytes)) return;
        JDoc d = JDoc::parse(bytes.data(), bytes.size());
        auto obj = [&](const char16_t* key) { int32_t m = d.member(d.root, key); return (m >= 0 && d.kind(m) == JKind::Object) ? m : -1; };
        FoldMap<W> al = aliases;
        FoldMap<std::vector<W>> im = implied;
        for (const char16_t* key : {u"aliases", u"normalization"}) {
            int32_t o = obj(key);
            if (o < 0) continue;
            for (int32_t c = d.nodes[o].first; c >= 0; c = d.nodes[c].next) al.put(normalize_basic(d.member_name(c)), normalize_basic(d.string_or_empty(c)));
        }
        if (int32_t o = obj(u"synonyms"); o >= 0) {
            for (int32_t c = d.nodes[o].first; c >= 0; c = d.nodes[c].next) {
                if (d.kind(c) != JKind::Array) continue;
                W canonical = normalize_basic(d.member_name(c));
                for (int32_t e = d.nodes[c].first; e >= 0; e = d.nodes[e].next) al.put(normalize_basic(d.string_or_empty(e)), canonical);
            }
        }
        for (const char16_t* key : {u"implied", u"implied_tags"}) {
            int32_t o = obj(key);
            if (o < 0) continue;
            for (int32_t c = d.nodes[o].first; c >= 0; c = d.nodes[c].next) im.put(normalize_basic(d.member_name(c)), string_array(d, c));
        }
        aliases = al;
        implied = im;
    }
    // ParseActionRule, Program.cs:1340-1372
    static bool parse_action(const JDoc& d, int32_t item, Action& out) {
        if (d.kind(item) != JKind::Object) return false;
        W parent;
        int32_t pt = d.member(item, u"parent_tag");
        if (pt >= 0) parent = d.string_or_empty(pt);
        if (blank(parent)) {
            int32_t pe = d.member(item, u"parent");
            if (pe >= 0) {
                if (d.kind(pe) == JKind::String) parent = d.str(pe);
                else if (d.kind(pe) == JKind::Array) {
                    parent.clear();
                    for (int32_t c = d.nodes[pe].first; c >= 0; c = d.nodes[c].next) {
                        W s = d.string_or_empty(c);
                        if (!blank(s)) { parent = s; break; }
                    }
                }
            }
        }
        if (blank(parent)) return false;
        out.all = opt_array(d, item, u"required_all");
        out.any = opt_array(d, item, u"required_any");
        out.forb = opt_array(d, item, u"forbidden_any");
        out.subs.clear();
        int32_t se = d.member(item, u"subtypes");
        if (se >= 0 && d.kind(se) == JKind::Array) {
            for (int32_t st = d.nodes[se].first; st >= 0; st = d.nodes[st].next) {
                if (d.kind(st) == JKind::String) {
                    W tag = normalize_basic(d.str(st));
                    if (!blank(tag)) out.subs.push_back(ActionSub{tag, {tag}, {}});
                } else if (d.kind(st) == JKind::Object) {
                    int32_t te = d.member(st, u"tag");
                    W tag = te >= 0 ? normalize_basic(d.string_or_empty(te)) : W();
                    if (blank(tag)) continue;
                    out.subs.push_back(ActionSub{tag, opt_array(d, st, u"match_any"), opt_array(d, st, u"match_all")});
                }
            }
        }
        out.parent = normalize_basic(parent);
        return true;
    }
    // LoadActions, Program.cs:1307-1338 (replaces the built-in rules when at least one parses)
    void load_actions(const tp_source& src) {
        std::vector<uint8_t> bytes;
        if (!read_source(src, bytes)) return;
        JDoc d = JDoc::parse(bytes.data(), bytes.size());
        std::vector<Action> list;
        if (d.kind(d.root) == JKind::Array) {
            for (int32_t c = d.nodes[d.root].first; c >= 0; c = d.nodes[c].next) { Action a; if (parse_action(d, c, a)) list.push_back(a); }
        } else if (d.kind(d.root) == JKind::Object) {
            if (d.member(d.root, u"parent_tag") >= 0 || d.member(d.root, u"parent") >= 0) {
                Action a;
                if (parse_action(d, d.root, a)) list.push_back(a);
            } else {
                for (int32_t c = d.nodes[d.root].first; c >= 0; c = d.nodes[c].next) {
                    if (d.kind(c) != JKind::Object) continue;
                    std::vector<W> parents = opt_array(d, c, u"parent"), subtypes = opt_array(d, c, u"subtypes");
                    for (auto& parent : parents) {
                        if (blank(parent)) continue;
                        Action a;
                        a.parent = parent;
                        a.all = {parent};
                        for (auto& st : subtypes) a.subs.push_back(ActionSub{st, {st}, {}});
                        list.push_back(a);
                    }
                }
            }
        }
        if (!list.empty()) actions = list;
    }
    // LoadRequiredFacets, Program.cs:1374-1406
    void load_facets(const tp_source& src) {
        std::vector<uint8_t> bytes;
        if (!read_source(src, bytes)) return;
        JDoc d = JDoc::parse(bytes.data(), bytes.size());
        if (d.kind(d.root) != JKind::Object) return;
        std::vector<FacetDef> list;
        W unknown_prefix = widen("unknown_");
        auto is_unknown = [&](const W& t) { return t.size() >= 8 && folded(t.substr(0, 8)) == folded(unknown_prefix); };
        for (int32_t c = d.nodes[d.root].first; c >= 0; c = d.nodes[c].next) {
            W name = normalize_basic(d.member_name(c));
            if (d.kind(c) == JKind::Array) {
                std::vector<W> tags = string_array(d, c);
                if (tags.empty()) continue;
                FacetDef f;
                f.name = name;
                auto it = std::find_if(tags.begin(), tags.end(), is_unknown);
                f.def = it != tags.end() ? *it : unknown_prefix + name;
                for (auto& t : tags) if (!is_unknown(t)) f.rules.push_back(FacetR{t, {t}, {}});
                list.push_back(f);
            } else if (d.kind(c) == JKind::Object) {
                FacetDef f;
                f.name = name;
                int32_t de = d.member(c, u"default");
                f.def = de >= 0 ? normalize_basic(d.string_or_empty(de)) : unknown_prefix + name;
                int32_t re = d.member(c, u"rules");
                if (re >= 0 && d.kind(re) == JKind::Array) {
                    for (int32_t r = d.nodes[re].first; r >= 0; r = d.nodes[r].next) {
                        if (d.kind(r) != JKind::Object) continue;
                        int32_t te = d.member(r, u"tag");
                        W tag = te >= 0 ? normalize_basic(d.string_or_empty(te)) : W();
                        if (blank(tag)) continue;
                        f.rules.push_back(FacetR{tag, opt_array(d, r, u"match_any"), opt_array(d, r, u"match_all")});
                    }
                }
                list.push_back(f);
            }
        }
        if (!list.empty()) facets = list;
    }
};

// Interned strings, unique under the ordinal-ignore-case comparison the reference's sets use.
struct Interner {
    std::vector<W> text;
    std::unordered_map<W, uint32_t> exact;
    std::unordered_map<W, uint32_t> by_fold;
    uint32_t id(const W& s) {
        auto it = exact.find(s);
        if (it != exact.end()) return it->second;
        W f = folded(s);
        auto jt = by_fold.find(f);
        if (jt != by_fold.end())
            throw ConfigError("two table strings differ only by case mapping (\"" + to_utf8(text[jt->second]) + "\" / \"" + to_utf8(s) + "\"): not representable");
        uint32_t n = (uint32_t)text.size();
        text.push_back(s);
        exact.emplace(s, n);
        by_fold.emplace(f, n);
        return n;
    }
};

}  // namespace

bool less_ordinal_ignore_case(const std::string& a, const std::string& b) {
    return folded(from_utf8((const uint8_t*)a.data(), a.size())) < folded(from_utf8((const uint8_t*)b.data(), b.size()));
}

void compile_tables(const tp_config& cfg, HostTables& out, const std::vector<std::string>* learned) {
    Loader ld;
    try {
        // ---- Main's load sequence, Program.cs:156-165 ----
        std::string trig = cfg.trigger ? cfg.trigger : "sgrui";
        ld.trigger = from_utf8((const uint8_t*)trig.data(), trig.size());
        ld.strict_explicit = cfg.strict_known_explicit != 0;
        ld.strict = cfg.strict_known != 0;
        std::string ap = cfg.auto_prefix_tags ? cfg.auto_prefix_tags : "", ms = cfg.minimal_subject_tags ? cfg.minimal_subject_tags : "";
        std::vector<W> hard_prefix = unique_fold(split_tags(from_utf8((const uint8_t*)ap.data(), ap.size())));   // :158
        ld.allow = ld.load_tag_file(cfg.allowlist);                                                                // :159
        ld.deny = ld.load_tag_file(cfg.denylist);
        ld.load_vocabulary(cfg);                                                                                   // :160
        if (!cfg.no_profile) ld.load_profile(cfg);                                                                 // :161
        ld.force_allow = FoldSet();                                                                                // BuildForceAllowTags, :933-940
        ld.force_allow.insert(ld.normalize(ld.trigger));
        ld.force_allow.insert(widen("1girl"));
        if (ld.has_profile) for (auto& t : ld.always_add) ld.force_allow.insert(ld.normalize(t));
        ld.default_lexicon();                                                                                      // LoadRuleEngineConfigs, :1212-1226
        ld.actions = Loader::builtin_actions();
        ld.facets = Loader::builtin_facets();
        ld.load_lexicon(cfg.lexicon);
        ld.load_actions(cfg.actions);
        ld.load_facets(cfg.required_facets);
        for (auto& a : ld.actions) { ld.action_tags.insert(a.parent); for (auto& s : a.subs) ld.action_tags.insert(s.tag); }   // RebuildActionTagSet, :1417-1422
        std::vector<W> mf = hard_prefix;                                                                           // :165
        for (auto& t : split_tags(from_utf8((const uint8_t*)ms.data(), ms.size()))) mf.push_back(t);
        mf = unique_fold(mf);
        W fallback_line;
        for (size_t i = 0; i < mf.size(); ++i) { if (i) fallback_line += widen(", "); fallback_line += mf[i]; }
        std::string fl8 = to_utf8(fallback_line);
        out.fallback_line.assign(fl8.begin(), fl8.end());
    } catch (const JsonFail& e) {
        throw ConfigError(std::string("Fatal error occurred. ") + e.what());
    }

    // ---- intern every string the pipeline can compare or emit; frequent ones first (low ids sit in the histogram's on-chip part) ----
    Interner in;
    in.id(W());                                                   // EMPTY_ID
    const W trigger_n = ld.normalize(ld.trigger);
    in.id(trigger_n);
    in.id(widen("1girl"));
    in.id(ld.normalize(widen("1girl")));
    std::vector<W> always_final;
    if (ld.has_profile) for (auto& t : ld.always_add) always_final.push_back(ld.normalize(ld.normalize(t)));   // Select(NormalizeTag) then AddOne's own NormalizeTag, Program.cs:667, 693
    if (always_final.size() > (size_t)MAX_ALWAYS_ADD) throw ConfigError("profile lists more than 64 always-add tags");
    for (auto& t : always_final) in.id(t);
    for (auto& f : ld.facets) { in.id(f.def); for (auto& r : f.rules) in.id(r.tag); }
    for (auto& a : ld.actions) { in.id(a.parent); for (auto& s : a.subs) in.id(s.tag); }
    for (auto& t : ld.known.order) in.id(t);
    for (auto& kv : ld.aliases.m) in.id(kv.second.first), in.id(kv.second.second);
    for (auto& kv : ld.builtin_aliases.m) in.id(kv.second.first), in.id(kv.second.second);
    for (auto& kv : ld.implied.m) { in.id(kv.second.first); for (auto& v : kv.second.second) in.id(ld.normalize(v)); }
    for (auto& a : ld.actions) {
        for (auto* l : {&a.all, &a.any, &a.forb}) for (auto& s : *l) in.id(s);
        for (auto& s : a.subs) for (auto* l : {&s.any, &s.all}) for (auto& x : *l) in.id(x);
    }
    for (auto& f : ld.facets) for (auto& r : f.rules) for (auto* l : {&r.any, &r.all}) for (auto& x : *l) in.id(x);
    for (auto* set : {&ld.subject, &ld.banned, &ld.builtin_deny, &ld.action_tags, &ld.safe_reject, &ld.allow, &ld.deny, &ld.force_allow})
        for (auto& s : set->order) in.id(s);
    const W feet_n = ld.normalize(widen("feet")), penis_n = ld.normalize(widen("penis")), footjob_n = ld.normalize(widen("footjob"));
    in.id(feet_n); in.id(penis_n); in.id(footjob_n);
    {
        W fl = from_utf8(out.fallback_line.data(), out.fallback_line.size());
        for (auto& tok : split_tags(fl)) in.id(normalize_basic(tok));
    }
    // A force-allowed tag survives ShouldDropTag whatever it holds (Program.cs:1123), so the line ingest keeps one that holds ','
    // (an alias value can be such a string) and its join + re-split (Program.cs:604, 578) cuts it into SplitTags' pieces, each a
    // candidate of its own from there on.  The pieces are interned as they stand; FL_SPLIT marks the string.
    std::vector<std::pair<W, std::vector<W>>> splits;
    for (auto& s : ld.force_allow.order) {
        if (s.size() < 2) continue;                               // Length >= 2 comes first (Program.cs:600)
        std::vector<W> pieces = split_tags(s);
        if (pieces.size() == 1 && pieces[0] == s) continue;
        if (splits.size() >= (size_t)MAX_SPLIT || pieces.size() > (size_t)MAX_SPLIT_PIECES)
            throw ConfigError("more than 4 force-allowed tags (trigger / always-add) hold ',', or one holds more than 7: not representable");
        for (auto& p : pieces) in.id(p);
        splits.emplace_back(s, pieces);
    }
    // close under NormalizeTagBasic, the alias lookup and NormalizeTag
    size_t closed = 0;
    auto close = [&]() {
        for (; closed < in.text.size(); ++closed) {
            W s = in.text[closed];
            in.id(ld.alias_of(s));
            W nb = normalize_basic(s);
            in.id(nb);
            in.id(ld.alias_of(nb));
        }
    };
    close();
    const uint32_t n_public = (uint32_t)in.text.size();
    if (learned) {
        for (const std::string& u : *learned) {
            const W wu = from_utf8((const uint8_t*)u.data(), u.size());
            if (in.exact.count(wu) || in.by_fold.count(folded(wu))) continue;
            if (blank(wu) || normalize_basic(wu) != wu) continue;     // only strings that normalisation leaves alone are learned
            in.id(wu);
        }
        close();
    }
    const uint32_t N = (uint32_t)in.text.size();


    // ---- per-string attributes ----
    out.strings.resize(N);
    out.str_off.assign(N + 1, 0);
    out.flags.assign(N, 0);
    out.next.assign(N, 0);
    out.basic_fixed.assign(N, 0);
    out.alias1.assign(N, 0);
    out.lit.assign(N, NO_ID);
    out.str_bytes.clear();
    auto id_of = [&](const W& s) -> uint32_t {
        auto it = in.exact.find(s);
        if (it != in.exact.end()) return it->second;
        auto jt = in.by_fold.find(folded(s));
        if (jt != in.by_fold.end()) return jt->second;
        throw ConfigError("internal: string missing from the intern table: " + to_utf8(s));
    };
    for (uint32_t i = 0; i < N; ++i) {
        const W& s = in.text[i];
        out.strings[i] = to_utf8(s);
        out.str_off[i] = (uint32_t)out.str_bytes.size();
        out.str_bytes.insert(out.str_bytes.end(), out.strings[i].begin(), out.strings[i].end());
        out.alias1[i] = id_of(ld.alias_of(s));
        out.next[i] = id_of(ld.alias_of(normalize_basic(s)));
        out.basic_fixed[i] = normalize_basic(s) == s ? 1 : 0;
        uint32_t mask = class_mask(ld.dfa_next, ld.dfa_out, lowered(s));
        uint32_t f = 0;
        if (blank(s)) f |= FL_BLANK;
        if (Loader::valid_tag(s)) f |= FL_VALID;
        if (ld.should_drop(s)) f |= FL_DROP;
        if (s.size() >= 2) f |= FL_LEN2;
        if (Loader::has_looks_bad_phrase(lowered(s))) f |= FL_PHRASE;
        if (s.find((wc)'`') != W::npos) f |= FL_FENCE;
        for (auto& sp : splits) if (folded(sp.first) == folded(s)) f |= FL_SPLIT;
        f |= ld.bucket_of(s, mask) << FL_BUCKET_SHIFT;
        if (mask & CM_HAIR) f |= FL_HAIR;
        if (mask & CM_EYES) f |= FL_EYES;
        if (ld.banned.contains(s)) f |= FL_BANNED;
        if (ld.safe_reject.contains(s)) {
            f |= FL_SAFE;
            W fs = folded(s);
            for (uint32_t k = 0; k < 20; ++k) if (folded(widen(kSafeList[k])) == fs) f |= k << FL_SAFE_IDX_SHIFT;
        }
        if (ld.known.contains(s)) f |= FL_KNOWN;
        {   // BuildReviewRecord's subject-token tests, Program.cs:829-837 (ordinal-ignore-case: on the folded string)
            const W fs = folded(s);
            auto holds = [&](const char* w) { return fs.find(widen(w)) != W::npos; };
            if (ld.subject.contains(s) || holds("BOYS") || holds("GIRLS")) {
                static const char* const kMale[8] = {"boy", "1boy", "2boys", "3boys", "4boys", "multiple_boys", "male_focus", "yaoi"};   // :30-32
                for (const char* m : kMale) if (fs == folded(widen(m))) f |= FL_SUBJ_MALE;
                if (holds("GIRL")) f |= FL_SUBJ_GIRL;
                bool multi = fs.size() >= 9 && fs.substr(0, 9) == widen("MULTIPLE_");
                size_t d = 0;
                auto nd = [](wc ch) {                               // \d: Unicode category Nd
                    if (ch >= '0' && ch <= '9') return true;
                    for (uint16_t z : kDigitZero) if (ch >= z && ch < z + 10) return true;
                    return false;
                };
                while (d < fs.size() && nd(fs[d])) ++d;
                if (d > 0) { const W r = fs.substr(d); multi |= r == widen("GIRL") || r == widen("GIRLS") || r == widen("BOY") || r == widen("BOYS"); }
                if (multi) f |= FL_SUBJ_MULTI;
            }
        }
        out.flags[i] = f;
    }
    out.str_off[N] = (uint32_t)out.str_bytes.size();
    out.str_bytes.resize(out.str_bytes.size() + 64, 0);   // slack: strings are fetched 32 bytes at a time whatever their length
    const uint32_t feet_id = id_of(feet_n), penis_id = id_of(penis_n), footjob_id = id_of(footjob_n);
    // implied lists (Program.cs:1171-1175): NormalizeTag of each entry, blanks removed
    out.imp_off.assign(N + 1, 0);
    out.imp_ids.clear();
    for (uint32_t i = 0; i < N; ++i) {
        out.imp_off[i] = (uint32_t)out.imp_ids.size();
        if (const std::vector<W>* lst = ld.implied.get(in.text[i]))
            for (auto& v : *lst) { W n = ld.normalize(v); if (!blank(n)) out.imp_ids.push_back(id_of(n)); }
    }
    out.imp_off[N] = (uint32_t)out.imp_ids.size();
    for (uint32_t i = 0; i < N; ++i) {
        if (out.next[i] != i || out.imp_off[i + 1] != out.imp_off[i]) out.flags[i] |= FL_EDGE;
        if (out.next[i] == feet_id) out.flags[i] |= FL_NFEET;
        if (out.next[i] == penis_id) out.flags[i] |= FL_NPENIS;
        if (out.next[i] == footjob_id) out.flags[i] |= FL_NFOOTJOB;
    }

    // ---- rules -> literal indices ----
    uint32_t n_lits = 0;
    auto lit_list = [&](const std::vector<W>& v, uint32_t& off, uint32_t& cnt) {
        off = (uint32_t)out.pool.size();
        cnt = (uint32_t)v.size();
        for (auto& s : v) {
            uint32_t id = id_of(s);
            if (out.lit[id] == NO_ID) out.lit[id] = n_lits++;
            out.pool.push_back(out.lit[id]);
        }
    };
    out.pool.clear(); out.rules.clear(); out.subs.clear(); out.facets.clear(); out.facet_rules.clear();
    out.safe_tags.assign(std::begin(kSafeList), std::end(kSafeList));
    out.rule_names.clear(); out.sub_names.clear(); out.facet_names.clear(); out.facet_defaults.clear(); out.facet_rule_tags.clear();
    for (auto& a : ld.actions) {
        DRule r{};
        r.parent_id = id_of(a.parent);
        lit_list(a.all, r.all_off, r.all_cnt);
        lit_list(a.any, r.any_off, r.any_cnt);
        lit_list(a.forb, r.forb_off, r.forb_cnt);
        r.sub_off = (uint32_t)out.subs.size();
        r.sub_cnt = (uint32_t)a.subs.size();
        for (auto& s : a.subs) {
            DSub d{};
            d.tag_id = id_of(s.tag);
            lit_list(s.any, d.any_off, d.any_cnt);
            lit_list(s.all, d.all_off, d.all_cnt);
            out.subs.push_back(d);
            out.sub_names.push_back(to_utf8(a.parent) + ":" + to_utf8(s.tag));
        }
        out.rules.push_back(r);
        out.rule_names.push_back(to_utf8(a.parent));
    }
    for (auto& f : ld.facets) {
        DFacet df{};
        df.def_id = id_of(f.def);
        df.rule_off = (uint32_t)out.facet_rules.size();
        df.rule_cnt = (uint32_t)f.rules.size();
        std::vector<std::string> tags;
        for (auto& r : f.rules) {
            DSub d{};
            d.tag_id = id_of(r.tag);
            lit_list(r.any, d.any_off, d.any_cnt);
            lit_list(r.all, d.all_off, d.all_cnt);
            out.facet_rules.push_back(d);
            tags.push_back(to_utf8(r.tag));
        }
        if (df.rule_cnt > 0xFFFE) throw ConfigError("a facet has more than 65534 rules");
        out.facets.push_back(df);
        out.facet_names.push_back(to_utf8(f.name));
        out.facet_defaults.push_back(to_utf8(f.def));
        out.facet_rule_tags.push_back(tags);
    }
    if (out.pool.empty()) out.pool.push_back(0);
    if (out.imp_ids.empty()) out.imp_ids.push_back(0);
    if (out.subs.empty()) out.subs.push_back(DSub{});
    if (out.rules.empty()) out.rules.push_back(DRule{});
    if (out.facets.empty()) out.facets.push_back(DFacet{});
    if (out.facet_rules.empty()) out.facet_rules.push_back(DSub{});

    // ---- hash table over the strings' bytes ----
    uint32_t slots = 16;
    while (slots < N * 8u + 8u) slots <<= 1;   // at most an eighth full: a wavefront probes until its slowest lane is done (measured: 7 % off the first wave stage against half full)
    out.ht_hash.assign(slots, 0);
    out.ht_id.assign(slots, NO_ID);
    uint64_t probe_sum = 0;
    uint32_t probe_max = 0, probe_multi = 0;
    for (uint32_t i = 0; i < N; ++i) {
        Hasher hs;
        for (unsigned char c : out.strings[i]) hs.add(c);
        uint64_t h = hs.done();
        uint32_t s = (uint32_t)h & (slots - 1);
        uint32_t probes = 1;
        while (out.ht_hash[s] != 0) { s = (s + 1) & (slots - 1); ++probes; }
        out.ht_hash[s] = h;
        out.ht_id[s] = i;
        probe_sum += probes;
        probe_max = std::max(probe_max, probes);
        if (probes > 1) ++probe_multi;
    }
    if (std::getenv("TAGPIPE_DEBUG_TABLES"))
        std::fprintf(stderr, "tagpipe tables: %u strings in %u slots, probes avg %.3f max %u, %.1f %% of the strings off their home slot\n", N, slots,
                     (double)probe_sum / std::max(1u, N), probe_max, 100.0 * probe_multi / std::max(1u, N));
    out.dfa_next = ld.dfa_next;
    out.dfa_out = ld.dfa_out;

    // ---- scalars ----
    DevTables& sc = out.scalars;
    std::memset(&sc, 0, sizeof sc);
    sc.n_strings = N;
    sc.n_public = n_public;
    sc.unk_keys = nullptr; sc.unk_ent = nullptr; sc.unk_text = nullptr; sc.unk_count = nullptr; sc.unk_mask = 0;
    sc.ht_mask = slots - 1;
    sc.n_lits = n_lits;
    sc.lit_words = (n_lits + 31) / 32;
    sc.n_rules = (uint32_t)ld.actions.size();
    sc.n_subs = (uint32_t)out.sub_names.size();
    sc.n_facets = (uint32_t)ld.facets.size();
    sc.fired_words = (sc.n_rules + sc.n_subs + 31) / 32;
    sc.trigger_id = id_of(trigger_n);
    sc.girl_add_id = id_of(ld.normalize(widen("1girl")));
    sc.girl_lit_id = id_of(widen("1girl"));
    sc.n_always = (uint32_t)always_final.size();
    for (uint32_t k = 0; k < sc.n_always; ++k) sc.always_ids[k] = id_of(always_final[k]);
    sc.n_fallback = 0;
    sc.feet_n = feet_id; sc.penis_n = penis_id; sc.footjob_n = footjob_id;
    {   // BannedFinalTags in the order OrderBy(OrdinalIgnoreCase) lists them (Program.cs:828): compared upper-cased
        static const char* const kBannedSorted[N_BANNED] = {"1boy", "2boys", "3boys", "4boys", "boy", "male", "male_focus", "multiple_boys", "yaoi"};
        for (int k = 0; k < N_BANNED; ++k) sc.banned_ids[k] = id_of(widen(kBannedSorted[k]));
    }
    sc.n_split = (uint32_t)splits.size();
    sc.split_grow = 1;
    for (uint32_t k = 0; k < sc.n_split; ++k) {
        sc.split_src[k] = id_of(splits[k].first);
        sc.split_cnt[k] = (uint32_t)splits[k].second.size();
        sc.split_commas[k] = (uint32_t)std::count(splits[k].first.begin(), splits[k].first.end(), (wc)',');
        for (uint32_t q = 0; q < sc.split_cnt[k]; ++q) sc.split_ids[k][q] = id_of(splits[k].second[q]);
        sc.split_grow = std::max(sc.split_grow, sc.split_cnt[k]);
    }
    sc.drop_unknown = (ld.known_found && ld.strict) ? 1 : 0;
    sc.allow_nonempty = ld.allow.size() > 0 ? 1 : 0;
    sc.known_active = (ld.known_found && ld.known.size() > 0) ? 1 : 0;
    sc.safe_mode = cfg.safe_mode ? 1 : 0;
    sc.line_weight = cfg.line_weight;
    sc.line_source = (uint32_t)cfg.line_source;
    std::string nl = cfg.newline ? cfg.newline : "\n";
    if (nl.size() > 4) throw ConfigError("newline longer than 4 bytes");
    sc.newline_len = (uint32_t)nl.size();
    for (size_t k = 0; k < nl.size(); ++k) sc.newline[k] = (uint8_t)nl[k];
    // Capacities of the per-file lists.  ExpandImpliedTags can add, per tag, everything reachable from it over the
    // NormalizeTag and implied edges; the largest such reach bounds the growth.  A file that still outgrows a list is
    // reported as unsupported rather than truncated.
    uint32_t reach_max = 0;
    {
        std::vector<uint32_t> seen(N, NO_ID), stack;
        for (uint32_t i = 0; i < N; ++i) {
            if (!(out.flags[i] & FL_EDGE)) continue;
            uint32_t reached = 0;
            stack.assign(1, i);
            seen[i] = i;
            while (!stack.empty() && reached < 4096) {
                uint32_t x = stack.back();
                stack.pop_back();
                auto visit = [&](uint32_t y) { if (seen[y] != i) { seen[y] = i; ++reached; stack.push_back(y); } };
                visit(out.next[x]);
                for (uint32_t m = out.imp_off[x]; m < out.imp_off[x + 1]; ++m) visit(out.imp_ids[m]);
            }
            reach_max = std::max(reach_max, reached);
        }
    }
    sc.final_cap = MAX_TOTAL_TAGS + std::min<uint32_t>(MAX_TOTAL_TAGS * reach_max, 2048) + sc.n_rules + sc.n_subs + sc.n_facets + 16;
    sc.seed_ci_limit = sc.style_ci_limit = 0;
    for (int ci = 0; ci < 4096; ++ci) {                             // the same arithmetic as the per-candidate score, Program.cs:581
        double score = std::fmax(0.05, 1.0 - (double)ci * 0.01) * cfg.line_weight;
        if (!(score < 0.28) && sc.seed_ci_limit == (uint32_t)ci) sc.seed_ci_limit = (uint32_t)ci + 1;
        if (score >= 0.60 && sc.style_ci_limit == (uint32_t)ci) sc.style_ci_limit = (uint32_t)ci + 1;
    }
    sc.evidence_grow = std::min<uint32_t>(73 * reach_max, 2048) + 16;   // at most 73 line-ingest candidates score >= 0.28
    // ---- tables of the wave-per-file kernel ----
    // prefix: the result list after AddOne(trigger), AddOne("1girl"), AddOne(always-add...) (Program.cs:664-668, 692-699)
    sc.n_prefix = 0;
    {
        std::vector<uint32_t> pre = {sc.trigger_id, sc.girl_add_id};
        for (uint32_t k = 0; k < sc.n_always; ++k) pre.push_back(sc.always_ids[k]);
        for (uint32_t id : pre) {
            uint32_t f = out.flags[id];
            if ((f & FL_BLANK) || sc.n_prefix >= (uint32_t)MAX_TOTAL_TAGS || !(f & FL_VALID) || (f & FL_DROP)) continue;
            if (std::find(sc.prefix_ids, sc.prefix_ids + sc.n_prefix, id) != sc.prefix_ids + sc.n_prefix) continue;
            sc.prefix_ids[sc.n_prefix++] = id;
            out.flags[id] |= FL_PREFIX;
        }
    }
    for (uint32_t i = 0; i < N; ++i) if (out.flags[out.next[i]] & FL_BLANK) out.flags[i] |= FL_NEXT_BLANK;
    for (uint32_t i = 0; i < N; ++i) if (out.next[i] == i && !(out.flags[i] & FL_BLANK)) out.flags[i] |= FL_FIX;
    // step lists: one ExpandImpliedTags visit of a string appends NormalizeTag(it), then that string's implied tags
    out.step_off.assign(N + 1, 0);
    out.step_ids.clear();
    for (uint32_t i = 0; i < N; ++i) {
        out.step_off[i] = (uint32_t)out.step_ids.size();
        if (!(out.flags[i] & FL_EDGE)) continue;
        uint32_t s = out.next[i];
        if (out.flags[s] & FL_BLANK) continue;                     // snapshot drops blank forms (Program.cs:1164)
        // a step that names the visited string itself, or repeats an earlier step of the same visit, can never append
        auto push = [&](uint32_t x) {
            if (x == i) return;
            for (uint32_t m = out.step_off[i]; m < out.step_ids.size(); ++m) if (out.step_ids[m] == x) return;
            out.step_ids.push_back(x);
        };
        push(s);
        for (uint32_t m = out.imp_off[s]; m < out.imp_off[s + 1]; ++m) push(out.imp_ids[m]);
    }
    out.step_off[N] = (uint32_t)out.step_ids.size();
    for (uint32_t i = 0; i < N; ++i) {                              // FL_EDGE now means exactly "a visit can append something"
        bool trivial = true;
        for (uint32_t m = out.step_off[i]; m < out.step_off[i + 1]; ++m) if (out.step_ids[m] != i) trivial = false;
        if (trivial) { out.flags[i] &= ~(uint32_t)FL_EDGE; }
    }
    // reach lists: closure of {it} under the visits, without it
    out.reach_off.assign(N + 1, 0);
    out.reach_ent.clear();
    sc.closure_tables = 1;
    {
        // Breadth first, in the order ExpandImpliedTags would grow the set {it}: round r appends, for every member that round r - 1
        // appended (in their order), what one visit of it appends (Program.cs:1160-1179).  An entry's round sits in bits 26..31 of
        // its info word, so that the lists of several seeds can be merged into the reference's insertion order.
        std::vector<uint32_t> seen(N, NO_ID), queue, level;
        sc.reach_levels_ok = 1;
        for (uint32_t i = 0; i < N && sc.closure_tables; ++i) {
            out.reach_off[i] = (uint32_t)(out.reach_ent.size() / 2);
            if (!(out.flags[i] & FL_EDGE)) continue;
            queue.assign(1, i);
            level.assign(1, 0);
            seen[i] = i;
            for (size_t q = 0; q < queue.size(); ++q) {
                const uint32_t x = queue[q], lv = level[q] + 1;
                if (!(out.flags[x] & FL_EDGE)) continue;
                for (uint32_t m = out.step_off[x]; m < out.step_off[x + 1]; ++m) {
                    uint32_t y = out.step_ids[m];
                    if (seen[y] == i) continue;
                    seen[y] = i;
                    queue.push_back(y);
                    level.push_back(lv);
                    if (lv > 63) sc.reach_levels_ok = 0;
                    uint32_t li = out.lit[y] == NO_ID ? 0xFFFFFFu : out.lit[y];
                    out.reach_ent.push_back(y);
                    out.reach_ent.push_back(li | ((out.flags[y] & FL_NFEET) ? 1u << 24 : 0) | ((out.flags[y] & FL_NPENIS) ? 1u << 25 : 0) | (std::min(lv, 63u) << 26));
                }
            }
            if (out.reach_ent.size() > (size_t)(64u << 20)) sc.closure_tables = 0;
        }
        if (!sc.closure_tables) { out.reach_ent.clear(); std::fill(out.reach_off.begin(), out.reach_off.end(), 0u); }
        out.reach_off[N] = (uint32_t)(out.reach_ent.size() / 2);
    }
    if (out.step_ids.empty()) out.step_ids.push_back(0);
    if (out.reach_ent.empty()) { out.reach_ent.push_back(0); out.reach_ent.push_back(0); }
    out.rec.resize(N);
    for (uint32_t i = 0; i < N; ++i) {
        uint32_t so = out.step_off[i], scnt = out.step_off[i + 1] - so;
        Rec r{};
        r.flags = out.flags[i]; r.next = out.next[i]; r.lit = out.lit[i]; r.len = out.str_off[i + 1] - out.str_off[i];
        r.len16 = (uint32_t)in.text[i].size();
        r.step_off = so; r.step_cnt = scnt;
        r.step0 = scnt > 0 ? out.step_ids[so] : NO_ID;
        r.step1 = scnt > 1 ? out.step_ids[so + 1] : NO_ID;
        r.s0_info = r.s1_info = 0xFFFFFFu;
        // FL_REACH2: the reach list is exactly the (at most two) steps, i.e. the steps lead nowhere further
        const uint32_t ro = out.reach_off[i], rn = out.reach_off[i + 1] - ro;
        if (sc.closure_tables && (out.flags[i] & FL_EDGE) && scnt <= 2 && rn == scnt) {
            bool same = true;
            for (uint32_t k = 0; k < scnt; ++k) if (out.reach_ent[2 * (ro + k)] != out.step_ids[so + k]) same = false;
            if (same) {
                r.flags |= FL_REACH2;
                out.flags[i] |= FL_REACH2;
                if (scnt > 0) r.s0_info = out.reach_ent[2 * ro + 1];
                if (scnt > 1) r.s1_info = out.reach_ent[2 * (ro + 1) + 1];
                bool leaf = true;
                for (uint32_t k = 0; k < scnt; ++k) {
                    const uint32_t y = out.step_ids[so + k];
                    if (out.next[y] != y || (out.flags[y] & FL_BLANK)) leaf = false;   // a step's own steps lie within the reach list, i.e. among these steps
                }
                if (leaf) {
                    r.flags |= FL_LEAF2;
                    out.flags[i] |= FL_LEAF2;
                    if (scnt > 0) { r.s0_flags = out.flags[r.step0]; r.s0_len = out.str_off[r.step0 + 1] - out.str_off[r.step0]; }
                    if (scnt > 1) { r.s1_flags = out.flags[r.step1]; r.s1_len = out.str_off[r.step1 + 1] - out.str_off[r.step1]; }
                }
            }
        }
        out.rec[i] = r;
    }
    std::vector<uint32_t> word_off(N);
    out.str_words.clear();
    for (uint32_t i = 0; i < N; ++i) {
        word_off[i] = (uint32_t)out.str_words.size();
        const std::string& str = out.strings[i];
        for (size_t b = 0; b < str.size(); b += 8) {
            uint64_t wv = 0;
            for (size_t k = 0; k < 8 && b + k < str.size(); ++k) wv |= (uint64_t)(unsigned char)str[b + k] << (8 * k);
            out.str_words.push_back(wv);
        }
    }
    for (int k = 0; k < 4; ++k) out.str_words.push_back(0);   // lookups compare four words at a time, whatever the string's length
    out.ht_info.assign(slots, HtInfo{0, 0, 0, 0});
    out.slots.assign(slots, Slot{});
    for (uint32_t s = 0; s < slots; ++s) {
        Slot& sl = out.slots[s];
        sl.slen = SLOT_EMPTY;
        if (!out.ht_hash[s]) continue;
        uint32_t id = out.ht_id[s];
        out.ht_info[s] = HtInfo{out.alias1[id], word_off[id], out.str_off[id + 1] - out.str_off[id], 0};
        sl.hash = (uint32_t)(out.ht_hash[s] >> 32);
        sl.slen = out.str_off[id + 1] - out.str_off[id];
        sl.val = out.alias1[id];
        sl.woff = word_off[id];
        sl.kflags = out.basic_fixed[id] ? SLOT_KEY_BASIC : 0;
        const std::string& str = out.strings[id];
        for (size_t b = 0; b < str.size() && b < 32; ++b) sl.key[b >> 2] |= (uint32_t)(unsigned char)str[b] << (8 * (b & 3));
        sl.rec = out.rec[sl.val];
    }
    {   // what ToLowerInvariant does outside ASCII, for the kernels (DevTables::case_map)
        std::map<wc, std::vector<wc>> by_capital;        // lower-case characters by the capital ordinal-ignore-case comparison folds them to
        for (uint32_t c = 0; c < 0x10000; ++c) if (lower_invariant((wc)c) == (wc)c) by_capital[upper_ordinal((wc)c)].push_back((wc)c);
        out.case_map.clear();
        for (uint32_t c = 0x80; c < 0x10000; ++c) {
            if (c >= 0xD800 && c <= 0xDFFF) continue;
            const wc l = lower_invariant((wc)c);
            if (l != (wc)c) out.case_map.push_back(c << 16 | l);
        }
        sc.n_case_map = (uint32_t)out.case_map.size();
        std::vector<uint32_t> twins;
        for (auto& kv : by_capital) if (kv.second.size() > 1) for (wc c : kv.second) if (c >= 0x80) twins.push_back(c);
        std::sort(twins.begin(), twins.end());
        sc.n_case_twin = (uint32_t)twins.size();
        out.case_map.insert(out.case_map.end(), twins.begin(), twins.end());
        sc.tables_hold_twins = 0;
        for (const W& w : in.text) for (wc c : w) if (c >= 0x80 && std::binary_search(twins.begin(), twins.end(), (uint32_t)c)) sc.tables_hold_twins = 1;
    }
    // ---- the lean stage's lookup table: two-choice placement (cuckoo insertion), at most 40 % full ----
    out.emit_rec.assign(4 * (size_t)N + 4, 0);
    for (uint32_t i = 0; i < N; ++i) {
        const std::string& str = out.strings[i];
        uint32_t* r = &out.emit_rec[4 * (size_t)i];
        if (str.size() <= 15) {
            uint8_t b[16] = {0};
            std::memcpy(b, str.data(), str.size());
            b[15] = (uint8_t)str.size();
            std::memcpy(r, b, 16);
        } else { r[0] = out.str_off[i]; r[1] = (uint32_t)str.size(); r[2] = 0; r[3] = 0xFFu << 24; }
    }
    out.emit_rec32.assign(8 * (size_t)N + 8, 0);
    for (uint32_t i = 0; i < N; ++i) {
        const std::string& str = out.strings[i];
        uint32_t* r = &out.emit_rec32[8 * (size_t)i];
        if (str.size() <= 31) {
            uint8_t b[32] = {0};
            std::memcpy(b, str.data(), str.size());
            b[31] = (uint8_t)str.size();
            std::memcpy(r, b, 32);
        } else { r[0] = out.str_off[i]; r[1] = (uint32_t)str.size(); r[7] = 0xFFu << 24; }
    }
    out.attr2.assign(2 * (size_t)N, 0);
    for (uint32_t i = 0; i < N; ++i) { out.attr2[2 * i] = out.flags[i]; out.attr2[2 * i + 1] = out.str_off[i + 1] - out.str_off[i]; }
    // final-list words and target indices (what can occur twice in a file's final list: steps, facet tags)
    out.fw_tix.assign(2 * (size_t)N, 0);
    {
        std::vector<uint8_t> target(N, 0);
        // (the lean stage's candidates are strings NormalizeTag leaves alone: only their steps can be among its list elements)
        for (uint32_t i = 0; i < N; ++i)
            if (out.flags[i] & FL_FIX) for (uint32_t m = out.step_off[i]; m < out.step_off[i + 1]; ++m) if (out.step_ids[m] < N) target[out.step_ids[m]] = 1;
        for (size_t f = 0; f < ld.facets.size(); ++f) {
            target[out.facets[f].def_id] = 1;
            for (uint32_t k = 0; k < out.facets[f].rule_cnt; ++k) target[out.facet_rules[out.facets[f].rule_off + k].tag_id] = 1;
        }
        uint32_t n_tix = 0;
        for (uint32_t i = 0; i < N; ++i) n_tix += target[i];
        sc.n_tix = n_tix;
        if (std::getenv("TAGPIPE_DEBUG_TABLES")) std::fprintf(stderr, "tagpipe tables: %u target indices (direct first-occurrence table up to %u)\n", n_tix, LEAN_TIX_MAX);
        uint32_t next_tix = 0;
        for (uint32_t i = 0; i < N; ++i) {
            const uint32_t fl = out.flags[i], len = out.str_off[i + 1] - out.str_off[i], bk = fl_bucket(fl);
            uint32_t fw = std::min<uint32_t>(len, FW_LEN_MASK);
            if (bk == B_FACE) fw |= FW_FACE;
            if (bk == B_OUTFIT) fw |= FW_OUTFIT;
            if (bk == B_POSE) fw |= FW_POSE;
            if (bk == B_BACKGROUND) fw |= FW_BACKGROUND;
            if (fl & FL_SAFE) fw |= FW_SAFE | ((fl >> FL_SAFE_IDX_SHIFT) & 31u) << FW_SAFE_IDX_SHIFT;
            if (i == sc.footjob_n) fw |= FW_FOOTJOB;
            if (!(fl & FL_KNOWN)) fw |= FW_UNKNOWN;
            if (i == sc.trigger_id) fw |= FW_TRIGGER;
            if (i == sc.girl_lit_id) fw |= FW_GIRL;
            if (fl & FL_HAIR) fw |= FW_HAIR;
            if (fl & FL_EYES) fw |= FW_EYES;
            if (fl & FL_BLANK) fw |= FW_BLANK;
            if (fl & FL_FIX) fw |= FW_FIX;
            if (fl & FL_EDGE) fw |= FW_EDGE;
            out.fw_tix[2 * (size_t)i] = fw;
            out.fw_tix[2 * (size_t)i + 1] = (target[i] && n_tix <= LEAN_TIX_MAX) ? next_tix++ : TIX_NONE;
        }
    }
    out.crec.assign(8 * (size_t)N, 0);
    for (uint32_t i = 0; i < N; ++i) {
        const Rec& rv = out.rec[i];
        uint32_t* cr = &out.crec[8 * (size_t)i];
        const bool edge = (rv.flags & FL_EDGE) != 0;
        const uint32_t step0 = edge ? rv.step0 : NO_ID, step1 = edge ? rv.step1 : NO_ID;
        auto len_of = [&](uint32_t id) { return id != NO_ID ? out.str_off[id + 1] - out.str_off[id] : 0u; };
        auto info16 = [](uint32_t info) {                           // Rec::s0_info in 16 bits: literal index (CREC_LIT_NONE = none), feet-ness, penis-ness
            const uint32_t li = info & 0xFFFFFFu;
            return (li == 0xFFFFFFu || li >= CREC_LIT_NONE ? CREC_LIT_NONE : li) | ((info >> 24) & 3u) << 12;
        };
        auto step_word = [&](uint32_t id) { return id == NO_ID || id > LEAN_HEAD_ID_MASK ? NO_ID : id | out.fw_tix[2 * (size_t)id + 1] << 20; };
        const bool too_long = rv.len >= FW_LEN_MASK || len_of(step0) >= FW_LEN_MASK || len_of(step1) >= FW_LEN_MASK;
        cr[0] = rv.flags;
        cr[1] = out.fw_tix[2 * (size_t)i];
        cr[2] = (rv.lit == NO_ID || rv.lit >= CREC_LIT_NONE ? CREC_LIT_NONE : rv.lit) | out.fw_tix[2 * (size_t)i + 1] << 12 |
                (edge ? std::min<uint32_t>(rv.step_cnt, 3) : 0u) << 24 | (too_long ? CREC_TOO_LONG : 0u);
        cr[3] = info16(step0 != NO_ID ? rv.s0_info : 0xFFFFFFu) | info16(step1 != NO_ID ? rv.s1_info : 0xFFFFFFu) << 16;
        cr[4] = step_word(step0); cr[5] = step_word(step1);
        cr[6] = step0 != NO_ID ? out.fw_tix[2 * (size_t)step0] : 0; cr[7] = step1 != NO_ID ? out.fw_tix[2 * (size_t)step1] : 0;
    }
    out.lslots.clear();
    sc.lslot_shift = 31;
    if (N <= LEAN_HEAD_ID_MASK) {
        std::vector<uint32_t> keys;
        for (uint32_t i = 0; i < N; ++i) {
            const uint32_t ln = out.str_off[i + 1] - out.str_off[i];
            if (out.basic_fixed[i] && ln >= 1 && ln <= LEAN_KEY_MAX) keys.push_back(i);
        }
        auto key_words = [&](uint32_t id, uint32_t kw[8]) {
            const std::string& str = out.strings[id];
            for (int k = 0; k < 8; ++k) kw[k] = 0;
            for (size_t b = 0; b < str.size(); ++b) kw[b >> 2] |= (uint32_t)(unsigned char)str[b] << (8 * (b & 3));
        };
        uint32_t lsl = 16;
        while ((uint64_t)lsl < (uint64_t)keys.size() * 2 + 8) lsl <<= 1;     // at most half full
        out.lkey_hi.assign(4, 0);
        for (int attempt = 0; attempt < 24 && out.lslots.empty(); ++attempt) {
            if (attempt && attempt % 8 == 0) lsl <<= 1;                     // eight seeds per size, then twice the slots
            const uint32_t seed = (uint32_t)attempt * 0x9E3779B9u;
            std::vector<uint32_t> owner(lsl, NO_ID);               // slot -> string id
            bool ok = true;
            for (uint32_t id0 : keys) {
                uint32_t id = id0, avoid = NO_ID;
                bool placed = false;
                for (int kick = 0; kick < 4000 && !placed; ++kick) {
                    uint32_t kw[8], s1, s2;
                    key_words(id, kw);
                    lean_slots(kw, (uint32_t)out.strings[id].size(), 32u - (uint32_t)__builtin_ctz(lsl), seed, s1, s2);
                    if (owner[s1] == NO_ID) { owner[s1] = id; placed = true; }
                    else if (owner[s2] == NO_ID) { owner[s2] = id; placed = true; }
                    else {                                          // evict: the slot it was not just thrown out of
                        const uint32_t victim_slot = (s1 == avoid) ? s2 : s1;
                        std::swap(id, owner[victim_slot]);
                        avoid = victim_slot;
                    }
                }
                if (!placed) { ok = false; if (std::getenv("TAGPIPE_DEBUG_TABLES")) std::fprintf(stderr, "tagpipe tables: lean slots attempt %d at %u slots: no place for \"%s\"\n", attempt, lsl, out.strings[id0].c_str()); break; }
            }
            if (!ok) continue;
            out.lslots.assign(lsl, LeanSlot{});
            for (uint32_t sidx = 0; sidx < lsl; ++sidx) {
                const uint32_t id = owner[sidx];
                if (id == NO_ID) continue;
                LeanSlot& ls = out.lslots[sidx];
                uint32_t kw[8];
                key_words(id, kw);
                for (int k = 0; k < 4; ++k) ls.key_lo[k] = kw[k];
                if (out.strings[id].size() > 16) {
                    ls.hi = (uint32_t)out.lkey_hi.size() / 4;
                    out.lkey_hi.insert(out.lkey_hi.end(), kw + 4, kw + 8);
                }
                const uint32_t val = out.alias1[id];
                const Rec& rv = out.rec[val];
                ls.head = val | ((uint32_t)out.strings[id].size() << 20) | (std::min<uint32_t>(rv.step_cnt, 3) << 26) | ((out.next[val] != val ? 1u : 0u) << 28);
                ls.flags = rv.flags;
            }
            sc.lslot_shift = 32u - (uint32_t)__builtin_ctz(lsl);
            sc.lslot_seed = seed;
        }
        if (std::getenv("TAGPIPE_DEBUG_TABLES")) std::fprintf(stderr, "tagpipe tables: lean slots %zu keys in %zu slots\n", keys.size(), out.lslots.size());
    }
    // rule cache
    bool rc_ok = true;
    {
        std::vector<uint32_t>& rc = out.rule_cache;
        rc.clear();
        std::vector<uint32_t> tail;                                // literal lists longer than four
        struct Fix { size_t word; uint32_t tail_off; };
        std::vector<Fix> fixes;
        auto record = [&](uint32_t tag, uint32_t w1, uint32_t sub_cnt, uint32_t all_off, uint32_t all_cnt, uint32_t any_off, uint32_t any_cnt, uint32_t forb_off,
                          uint32_t forb_cnt) {
            if (all_cnt > 255 || any_cnt > 255 || forb_cnt > 255 || sub_cnt > 0xFFFF) rc_ok = false;
            size_t base = rc.size();
            rc.resize(base + RC_REC_WORDS, 0);
            rc[base] = tag;
            rc[base + 1] = w1;
            rc[base + 2] = (all_cnt & 255) | ((any_cnt & 255) << 8) | ((forb_cnt & 255) << 16);
            rc[base + 3] = sub_cnt << 16;
            rc[base + 8] = out.flags[tag];
            rc[base + 9] = out.str_off[tag + 1] - out.str_off[tag];
            rc[base + 10] = out.fw_tix[2 * (size_t)tag];
            rc[base + 11] = out.fw_tix[2 * (size_t)tag + 1];
            if (out.next[tag] != tag || (out.flags[tag] & FL_BLANK)) rc[base + 3] |= 0x8000u;   // NormalizeTag does not leave this tag alone
            std::vector<uint32_t> lits;
            for (uint32_t k = 0; k < all_cnt; ++k) lits.push_back(out.pool[all_off + k]);
            for (uint32_t k = 0; k < any_cnt; ++k) lits.push_back(out.pool[any_off + k]);
            for (uint32_t k = 0; k < forb_cnt; ++k) lits.push_back(out.pool[forb_off + k]);
            if (lits.size() <= 4) {
                for (size_t k = 0; k < lits.size(); ++k) rc[base + 4 + k] = lits[k];
            } else {
                fixes.push_back(Fix{base + 3, (uint32_t)tail.size()});
                tail.insert(tail.end(), lits.begin(), lits.end());
            }
        };
        const uint32_t n_rules = (uint32_t)ld.actions.size(), n_subs = (uint32_t)out.sub_names.size(), n_facets = (uint32_t)ld.facets.size();
        for (uint32_t r = 0; r < n_rules; ++r) {
            const DRule& d = out.rules[r];
            record(d.parent_id, d.sub_off, d.sub_cnt, d.all_off, d.all_cnt, d.any_off, d.any_cnt, d.forb_off, d.forb_cnt);
        }
        sc.rc_subs = (uint32_t)rc.size();
        for (uint32_t k = 0; k < n_subs; ++k) {
            const DSub& d = out.subs[k];
            record(d.tag_id, 0, 0, d.all_off, d.all_cnt, d.any_off, d.any_cnt, 0, 0);
        }
        sc.rc_frules = (uint32_t)rc.size();
        uint32_t n_frules = 0;
        for (uint32_t f = 0; f < n_facets; ++f) n_frules += out.facets[f].rule_cnt;
        for (uint32_t k = 0; k < n_frules; ++k) {
            const DSub& d = out.facet_rules[k];
            record(d.tag_id, 0, 0, d.all_off, d.all_cnt, d.any_off, d.any_cnt, 0, 0);
        }
        sc.n_frules = n_frules;
        sc.rc_facets = (uint32_t)rc.size();
        for (uint32_t f = 0; f < n_facets; ++f) {
            const uint32_t d = out.facets[f].def_id;
            rc.push_back(d); rc.push_back(out.facets[f].rule_off); rc.push_back(out.facets[f].rule_cnt);
            rc.push_back(out.flags[d]); rc.push_back(out.str_off[d + 1] - out.str_off[d]);
            rc.push_back((out.next[d] != d || (out.flags[d] & FL_BLANK)) ? 1u : 0u);
            rc.push_back(out.fw_tix[2 * (size_t)d]); rc.push_back(out.fw_tix[2 * (size_t)d + 1]);
        }
        {   // positive-literal masks
            std::vector<uint32_t> rmask(sc.lit_words, 0), fmask(sc.lit_words, 0);
            sc.rules_vacuous = sc.frules_vacuous = 0;
            auto mark = [&](std::vector<uint32_t>& m, uint32_t off, uint32_t cnt) { for (uint32_t k = 0; k < cnt; ++k) { uint32_t li = out.pool[off + k]; m[li >> 5] |= 1u << (li & 31); } };
            // An action rule with required literals fires only when all of them are in the evidence: one of them per rule is
            // enough for the mask, and the one most rules share keeps the mask small (the built-in rules all require "penis").
            std::unordered_map<uint32_t, uint32_t> all_uses;
            for (uint32_t r = 0; r < n_rules; ++r) for (uint32_t k = 0; k < out.rules[r].all_cnt; ++k) ++all_uses[out.pool[out.rules[r].all_off + k]];
            for (uint32_t r = 0; r < n_rules; ++r) {
                const DRule& d = out.rules[r];
                if (d.all_cnt + d.any_cnt == 0) sc.rules_vacuous = 1;
                if (d.all_cnt) {
                    uint32_t best = out.pool[d.all_off];
                    for (uint32_t k = 1; k < d.all_cnt; ++k) { const uint32_t li = out.pool[d.all_off + k]; if (all_uses[li] > all_uses[best]) best = li; }
                    rmask[best >> 5] |= 1u << (best & 31);
                } else mark(rmask, d.any_off, d.any_cnt);
            }
            for (uint32_t k = 0; k < n_frules; ++k) {
                const DSub& d = out.facet_rules[k];
                if (d.all_cnt + d.any_cnt == 0) sc.frules_vacuous = 1;
                mark(fmask, d.all_off, d.all_cnt); mark(fmask, d.any_off, d.any_cnt);
            }
            sc.rc_rule_mask = (uint32_t)rc.size();
            rc.insert(rc.end(), rmask.begin(), rmask.end());
            sc.rc_frule_mask = (uint32_t)rc.size();
            rc.insert(rc.end(), fmask.begin(), fmask.end());
        }
        const uint32_t tail_base = (uint32_t)rc.size();
        for (const Fix& fx : fixes) rc[fx.word] |= (tail_base + fx.tail_off) & 0x7FFFu;
        rc.insert(rc.end(), tail.begin(), tail.end());
        if (rc.empty()) rc.push_back(0);
        sc.rc_words = (uint32_t)rc.size();
        sc.prefix_simple = 1;
        for (uint32_t k = 0; k < sc.n_prefix; ++k) {
            const uint32_t id = sc.prefix_ids[k];
            sc.prefix_flags[k] = out.flags[id];
            sc.prefix_len[k] = out.str_off[id + 1] - out.str_off[id];
            if (out.next[id] != id || (out.flags[id] & (FL_BLANK | FL_EDGE))) sc.prefix_simple = 0;
        }
        if (std::getenv("TAGPIPE_DEBUG_TABLES")) {
            uint32_t n_edge = 0, n_leaf = 0, n_reach2 = 0;
            for (uint32_t i = 0; i < N; ++i) { n_edge += (out.flags[i] & FL_EDGE) != 0; n_leaf += (out.flags[i] & FL_LEAF2) != 0; n_reach2 += (out.flags[i] & FL_REACH2) != 0; }
            std::fprintf(stderr, "tagpipe tables: strings %u edge %u reach2 %u leaf2 %u prefix_simple %u rc_words %u\n", N, n_edge, n_reach2, n_leaf, sc.prefix_simple,
                         sc.rc_words);
        }
        if (sc.rc_words > FAST_RC_MAX_WORDS || sc.rc_words > 0x7FFFu || n_frules > FAST_FRULES_MAX) rc_ok = false;
    }
    // the wave-per-file kernel keeps its literal bitset in 128 LDS words and assumes scores fall with the token index
    sc.fast_ok = (sc.lit_words <= 128 && sc.fired_words <= 64 && n_lits < 0xFFFFFFu && cfg.line_weight > 0.0 && rc_ok) ? 1 : 0;
    sc.n_fb_final = 0;
    out.n_known = (uint32_t)ld.known.size();
    out.strict = ld.strict;
    out.known_found = ld.known_found;
}

void set_fallback(const HostTables& ht, DevTables& dt, const uint32_t* ids, uint32_t n) {
    if (n > (uint32_t)MAX_FALLBACK) throw ConfigError("internal: fallback selection too long");
    dt.n_fallback = n;
    dt.n_fb_final = 0;
    for (uint32_t i = 0; i < n; ++i) {
        dt.fallback_ids[i] = ids[i];
        uint32_t h = ht.next[ids[i]];                               // finalTags.Select(NormalizeTag).Where(non-blank).ToHashSet(), Program.cs:303
        if (ht.flags[h] & FL_BLANK) continue;
        if (std::find(dt.fb_final, dt.fb_final + dt.n_fb_final, h) != dt.fb_final + dt.n_fb_final) continue;
        dt.fb_final[dt.n_fb_final++] = h;
    }
}

std::string top_tags_text(const HostTables& h, const uint64_t* histogram) {
    std::vector<uint32_t> ids;
    for (uint32_t i = 0; i < h.scalars.n_public; ++i) if (histogram[i]) ids.push_back(i);
    std::sort(ids.begin(), ids.end(), [&](uint32_t a, uint32_t b) {
        if (histogram[a] != histogram[b]) return histogram[a] > histogram[b];
        return h.strings[a] < h.strings[b];
    });
    std::string text;
    for (uint32_t i : ids) { text += h.strings[i]; text += '\n'; }
    return text;
}

std::unordered_map<std::string, uint32_t> index_strings(const HostTables& ht) {
    std::unordered_map<std::string, uint32_t> m;
    m.reserve(ht.strings.size() * 2);
    for (uint32_t i = 0; i < (uint32_t)ht.strings.size(); ++i) m.emplace(ht.strings[i], i);
    return m;
}

void decode_joytag_reply(const HostTables& ht, const std::unordered_map<std::string, uint32_t>& string_ids, const uint8_t* json, size_t len,
                         DecodedReplies& out, uint8_t& status, uint8_t& has_score) {
    status = TP_REPLY_OK;
    has_score = 0;
    struct Item { W tag; double score; };
    std::vector<Item> items;
    try {
        JDoc d = JDoc::parse(json, len);
        const int32_t tags = d.member(d.root, u"tags");                       // TryGetProperty throws unless the root is an object
        if (tags < 0 || d.kind(tags) != JKind::Array) { status = TP_REPLY_NO_TAGS; return; }
        // NormalizeTag(tag) is blank?  Only tags that are not blank take an Order (Program.cs:470, 484), and a string element's
        // score falls with that Order.
        auto blank = [&](const W& raw) {
            const W nb = normalize_basic(raw);
            auto it = string_ids.find(to_utf8(nb));
            if (it == string_ids.end()) return nb.empty();                     // in no table: no alias applies, and white space has become '_'
            return (ht.flags[ht.alias1[it->second]] & FL_BLANK) != 0;
        };
        auto number = [&](int32_t n, double& v) {                              // JsonElement.TryGetDouble: throws unless the element is a number
            if (d.kind(n) != JKind::Number) throw JsonFail("a JSON number was expected");
            v = std::strtod(to_utf8(d.str(n)).c_str(), nullptr);
            return true;
        };
        int order = 0;
        for (int32_t e = d.nodes[tags].first; e >= 0; e = d.nodes[e].next) {
            if (d.kind(e) == JKind::String) {
                const W& raw = d.str(e);
                if (!blank(raw)) { items.push_back(Item{raw, std::max(0.01, 1.0 - order * 0.001)}); ++order; }
                continue;
            }
            if (d.kind(e) != JKind::Object) continue;
            W tag;
            int32_t m = d.member(e, u"tag");
            if (m >= 0 && d.kind(m) == JKind::String) tag = d.str(m);
            else if ((m = d.member(e, u"name")) >= 0 && d.kind(m) == JKind::String) tag = d.str(m);
            double score = 0.5;
            if ((m = d.member(e, u"score")) >= 0 && number(m, score)) has_score = 1;
            else if ((m = d.member(e, u"confidence")) >= 0 && number(m, score)) has_score = 1;
            if (!blank(tag)) { items.push_back(Item{tag, score}); ++order; }
        }
    } catch (const JsonFail&) {
        status = TP_REPLY_FAILED;
        has_score = 0;
        return;
    }
    // With HasScore the list is sorted by (score descending, order) (Program.cs:487), which is what the scored ingest does with any
    // list.  Without it the list stays as it is: that is the same thing only when the scores do not rise along it; the status
    // tells the caller when they do (tp_run_replies keeps such a list's order, tp_run_scored / tp_run_merged would sort it).
    if (!has_score)
        for (size_t k = 1; k < items.size(); ++k) if (items[k].score > items[k - 1].score) { status = TP_REPLY_ORDER; break; }
    for (const Item& it : items) {
        const std::string b = to_utf8(it.tag);
        out.arena.insert(out.arena.end(), b.begin(), b.end());
        out.tag_offsets.push_back(out.arena.size());
        out.scores.push_back(it.score);
    }
}

DevTables bind_host_pointers(const HostTables& ht) {
    DevTables t = ht.scalars;
    t.str_bytes = ht.str_bytes.data();
    t.str_off = ht.str_off.data();
    t.flags = ht.flags.data();
    t.next = ht.next.data();
    t.alias1 = ht.alias1.data();
    t.lit = ht.lit.data();
    t.imp_off = ht.imp_off.data();
    t.imp_ids = ht.imp_ids.data();
    t.ht_hash = ht.ht_hash.data();
    t.ht_id = ht.ht_id.data();
    t.dfa_next = ht.dfa_next.data();
    t.dfa_out = ht.dfa_out.data();
    t.pool = ht.pool.data();
    t.rules = ht.rules.data();
    t.subs = ht.subs.data();
    t.facets = ht.facets.data();
    t.facet_rules = ht.facet_rules.data();
    t.rec = ht.rec.data();
    t.ht_info = ht.ht_info.data();
    t.str_words = ht.str_words.data();
    t.rc = ht.rule_cache.data();
    t.slots = ht.slots.data();
    t.lslots = ht.lslots.empty() ? nullptr : ht.lslots.data();
    t.lkey_hi = ht.lkey_hi.data();
    t.attr2 = ht.attr2.data();
    t.emit_rec = ht.emit_rec.data();
    t.emit_rec32 = ht.emit_rec32.data();
    t.case_map = ht.case_map.data();
    t.crec = ht.crec.data();
    t.fw_tix = ht.fw_tix.data();
    t.step_off = ht.step_off.data();
    t.step_ids = ht.step_ids.data();
    t.reach_off = ht.reach_off.data();
    t.reach_ent = ht.reach_ent.data();
    return t;
}

}  // namespace tp

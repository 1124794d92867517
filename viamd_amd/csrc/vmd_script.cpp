// Mini script front-end in C++: md_script_ir_compile_from_source stand-in (/root/reference/src/main.cpp:878) for
// the subset of the VIAMD script language the hot path needs - the forms VIAMD ships as defaults or generates itself
// (src/main.cpp:528, :2817-2858):
//
//     s1 = resname("ALA")[2:8];
//     r  = rdf(element('C'), element('H'), 10.0);            # also rdf(a, b, {rmin, rmax}) and rdf(a, b, rmin:rmax)
//     v  = sdf(s1, element('H'), 10.0);
//     d1 = distance(10, 30);                                  # 1-based atom indices (src/main.cpp:2817)
//     d2 = distance_min(1:3, element('O')) in residue(2:9);   # one value per context, indices local to the context
//
// Selections: element('X') | type/name/label('X') | resname("X") | residue(a:b) | resid(a:b) | atom(a:b) | integer [: integer] |
// all | water | protein | identifiers bound earlier, combined with `and`, `or`, `not`, parentheses; `sel[a:b]` slices an array of
// structures (1-based, inclusive).  Output: property descriptors appended to a vmd_script_ir_t (vmd_ir_add_*); nothing is
// evaluated here.  viamd_amd/script.py is the same front-end in Python; tests/test_script.py checks that both produce
// identical IR fingerprints.
#include <algorithm>
#include <cctype>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <map>
#include <memory>
#include <set>
#include <stdexcept>
#include <string>
#include <vector>

#include "vmd_eval.h"

extern "C" void vmd_set_last_error(const char* msg);

namespace {

struct ScriptError : std::runtime_error { using std::runtime_error::runtime_error; };

[[noreturn]] void fail(const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    throw ScriptError(buf);
}

enum TokKind { T_END, T_NUM, T_STR, T_ID, T_OP };
struct Token { TokKind kind; std::string text; size_t beg = 0, end = 0; };     // [beg, end): byte range in the source

// tolerant: a character outside the subset becomes a one-character T_OP token instead of an error (partial compilation: the statement
// it belongs to is then reported as skipped, the others still compile)
std::vector<Token> tokenize(const char* src, bool tolerant = false) {
    std::string text(src);
    // strip comments
    for (size_t i = 0; i < text.size(); ++i)
        if (text[i] == '#') { size_t j = i; while (j < text.size() && text[j] != '\n') text[j++] = ' '; }
    std::vector<Token> out;
    size_t p = 0;
    const size_t n = text.size();
    while (p < n) {
        if (isspace((unsigned char)text[p])) { ++p; continue; }
        const char c = text[p];
        if (isdigit((unsigned char)c) || (c == '.' && p + 1 < n && isdigit((unsigned char)text[p + 1]))) {
            size_t q = p;
            while (q < n && isdigit((unsigned char)text[q])) ++q;
            if (q < n && text[q] == '.') {
                ++q;
                while (q < n && isdigit((unsigned char)text[q])) ++q;
                if (q < n && (text[q] == 'e' || text[q] == 'E')) {
                    size_t r = q + 1;
                    if (r < n && (text[r] == '+' || text[r] == '-')) ++r;
                    if (r < n && isdigit((unsigned char)text[r])) { while (r < n && isdigit((unsigned char)text[r])) ++r; q = r; }
                }
            }
            out.push_back({T_NUM, text.substr(p, q - p), p, q});
            p = q;
        } else if (c == '\'' || c == '"') {
            const size_t q = text.find(c, p + 1);
            if (q == std::string::npos) {
                if (!tolerant) fail("unexpected character '%c' at offset %zu", c, p);
                out.push_back({T_OP, std::string(1, c), p, p + 1});
                ++p;
                continue;
            }
            out.push_back({T_STR, text.substr(p + 1, q - p - 1), p, q + 1});
            p = q + 1;
        } else if (isalpha((unsigned char)c) || c == '_') {
            size_t q = p;
            while (q < n && (isalnum((unsigned char)text[q]) || text[q] == '_')) ++q;
            out.push_back({T_ID, text.substr(p, q - p), p, q});
            p = q;
        } else if (strchr("=(),;:[]{}", c) || tolerant) {
            out.push_back({T_OP, std::string(1, c), p, p + 1});
            ++p;
        } else {
            fail("unexpected character '%c' at offset %zu", c, p);
        }
    }
    return out;
}

struct Topo {
    size_t n = 0, nres = 0;
    std::vector<std::string> elements, names, resnames;
    std::vector<std::vector<int32_t>> res_atoms;     // ascending atom indices per residue
    std::vector<int32_t> res_seq;                    // residue sequence number of the file per residue (resid()); empty = unknown

    explicit Topo(const vmd_topology_t* t) {
        n = t->num_atoms;
        elements.resize(n); names.resize(n); resnames.resize(n);
        std::vector<int32_t> ri(n, 0);
        for (size_t i = 0; i < n; ++i) {
            elements[i] = t->elements && t->elements[i] ? t->elements[i] : "";
            names[i] = t->names && t->names[i] ? t->names[i] : elements[i];
            resnames[i] = t->resnames && t->resnames[i] ? t->resnames[i] : "UNK";
            ri[i] = t->residue_index ? t->residue_index[i] : 0;
            if (ri[i] < 0) fail("topology: negative residue index at atom %zu", i);
            nres = std::max(nres, (size_t)ri[i] + 1);
        }
        if (n == 0) nres = 0;
        res_atoms.resize(nres);
        for (size_t i = 0; i < n; ++i) res_atoms[ri[i]].push_back((int32_t)i);
        if (t->residue_seq_id) {
            res_seq.assign(nres, 0);
            for (size_t r = 0; r < nres; ++r) if (!res_atoms[r].empty()) res_seq[r] = t->residue_seq_id[res_atoms[r][0]];
        }
    }
    const std::string& residue_name(size_t r) const { static const std::string unk = "UNK"; return res_atoms[r].empty() ? unk : resnames[res_atoms[r][0]]; }
};

std::string upper(std::string s) { for (auto& c : s) c = (char)toupper((unsigned char)c); return s; }
const std::set<std::string> WATER = {"HOH", "WAT", "SOL", "TIP3", "TIP4", "SPC", "H2O"};
const std::set<std::string> PROTEIN = {"ALA", "ARG", "ASN", "ASP", "CYS", "GLN", "GLU", "GLY", "HIS", "ILE", "LEU", "LYS", "MET", "PHE",
                                       "PRO", "SER", "THR", "TRP", "TYR", "VAL"};

struct Sel {
    std::vector<uint8_t> mask;
    bool has_structs = false;
    std::vector<std::vector<int32_t>> structs;
    std::vector<int32_t> indices() const {
        std::vector<int32_t> out;
        for (size_t i = 0; i < mask.size(); ++i) if (mask[i]) out.push_back((int32_t)i);
        return out;
    }
};

struct Parser {
    const std::vector<Token>& t;
    size_t i = 0;
    const Topo& topo;
    const std::map<std::string, Sel>& env;
    const std::vector<int32_t>* ctx;          // atoms of the evaluation context (`... in residue(3)`), or null
    std::vector<uint8_t> ctx_mask;

    Parser(const std::vector<Token>& toks, const Topo& tp, const std::map<std::string, Sel>& e, const std::vector<int32_t>* c = nullptr)
        : t(toks), topo(tp), env(e), ctx(c) {
        if (ctx) { ctx_mask.assign(topo.n, 0); for (int32_t a : *ctx) ctx_mask[a] = 1; }
    }
    const Token& peek() const { static const Token end{T_END, ""}; return i < t.size() ? t[i] : end; }
    bool is_word(const char* w) const { const Token& k = peek(); return (k.kind == T_ID || k.kind == T_OP) && k.text == w; }
    std::string take(const char* value, TokKind kind = T_END) {
        const Token& k = peek();
        if (k.kind == T_END || (value && k.text != value) || (kind != T_END && k.kind != kind))
            fail("expected %s, found '%s'", value ? value : (kind == T_NUM ? "num" : kind == T_STR ? "str" : "id"), k.text.c_str());
        ++i;
        return k.text;
    }
    bool accept(const char* value) { if (is_word(value)) { ++i; return true; } return false; }

    long integer() {
        const std::string s = take(nullptr, T_NUM);
        for (char c : s) if (!isdigit((unsigned char)c)) fail("expected an integer, found '%s'", s.c_str());
        return strtol(s.c_str(), nullptr, 10);
    }
    // range a[:b], 1-based inclusive -> 0-based [a, b)
    void range(long& a, long& b) {
        a = integer(); b = a;
        if (accept(":")) b = integer();
        if (a < 1 || b < a) fail("bad range %ld:%ld (script indices are 1-based)", a, b);
        a -= 1;
    }
    double number() { return strtod(take(nullptr, T_NUM).c_str(), nullptr); }

    Sel sel_or() {
        Sel s = sel_and();
        while (accept("or")) {
            const Sel r = sel_and();
            Sel o; o.mask.resize(topo.n);
            for (size_t k = 0; k < topo.n; ++k) o.mask[k] = s.mask[k] | r.mask[k];
            s = o;
        }
        return s;
    }
    Sel sel_and() {
        Sel s = sel_not();
        while (accept("and")) {
            const Sel r = sel_not();
            Sel o; o.mask.resize(topo.n);
            for (size_t k = 0; k < topo.n; ++k) o.mask[k] = s.mask[k] & r.mask[k];
            if (s.has_structs) {      // per-structure intersection keeps the array shape
                o.has_structs = true;
                for (auto& st : s.structs) { std::vector<int32_t> f; for (int32_t a : st) if (r.mask[a]) f.push_back(a); o.structs.push_back(f); }
            }
            s = o;
        }
        return s;
    }
    Sel sel_not() {
        if (accept("not")) {
            const Sel r = sel_not();
            Sel o; o.mask.resize(topo.n);
            for (size_t k = 0; k < topo.n; ++k) o.mask[k] = !r.mask[k];
            return o;
        }
        return sel_postfix();
    }
    Sel from_structs(const std::vector<std::vector<int32_t>>& st) {
        Sel s; s.mask.assign(topo.n, 0); s.has_structs = true; s.structs = st;
        for (auto& x : st) for (int32_t a : x) s.mask[a] = 1;
        return s;
    }
    Sel sel_postfix() {
        Sel s = sel_atom();
        while (accept("[")) {
            long a, b;
            range(a, b);
            take("]");
            if (!s.has_structs) fail("[a:b] applies to an array of structures (resname(...), residue(...))");
            if ((size_t)b > s.structs.size()) fail("slice [%ld:%ld] exceeds the %zu structures of the selection", a + 1, b, s.structs.size());
            s = from_structs(std::vector<std::vector<int32_t>>(s.structs.begin() + a, s.structs.begin() + b));
        }
        return s;
    }
    template <class Pred> Sel residues(Pred pred) {
        std::vector<std::vector<int32_t>> st;
        for (size_t r = 0; r < topo.nres; ++r) if (pred(r)) st.push_back(topo.res_atoms[r]);
        return from_structs(st);
    }
    Sel atom_range(long a, long b) {
        Sel s; s.mask.assign(topo.n, 0);
        if (ctx) {
            if ((size_t)b > ctx->size()) fail("atom index %ld out of range (the context has %zu atoms)", b, ctx->size());
            for (long k = a; k < b; ++k) s.mask[(*ctx)[k]] = 1;
        } else {
            if ((size_t)b > topo.n) fail("atom index %ld out of range (system has %zu atoms)", b, topo.n);
            for (long k = a; k < b; ++k) s.mask[k] = 1;
        }
        return s;
    }
    Sel sel_atom() {
        Sel s = sel_atom_raw();
        if (ctx && !is_word("[")) {
            for (size_t k = 0; k < topo.n; ++k) s.mask[k] &= ctx_mask[k];
            if (s.has_structs) for (auto& st : s.structs) { std::vector<int32_t> f; for (int32_t a : st) if (ctx_mask[a]) f.push_back(a); st = f; }
        }
        return s;
    }
    Sel sel_atom_raw() {
        const Token k = peek();
        if (k.kind == T_OP && k.text == "(") {
            take("(");
            Sel s = sel_or();
            take(")");
            return s;
        }
        if (k.kind == T_NUM) {
            long a, b;
            range(a, b);
            return atom_range(a, b);
        }
        if (k.kind != T_ID) fail("unexpected token '%s' in selection", k.text.c_str());
        ++i;
        const std::string& v = k.text;
        auto it = env.find(v);
        if (it != env.end()) return it->second;
        if (v == "all") { Sel s; s.mask.assign(topo.n, 1); return s; }
        if (v == "water") return residues([&](size_t r) { return WATER.count(upper(topo.residue_name(r))) != 0; });
        if (v == "protein") return residues([&](size_t r) { return PROTEIN.count(upper(topo.residue_name(r))) != 0; });
        if (v == "element" || v == "type" || v == "name" || v == "label" || v == "resname") {
            take("(");
            std::vector<std::string> names{take(nullptr, T_STR)};
            while (accept(",")) names.push_back(take(nullptr, T_STR));
            take(")");
            auto in_names = [&](const std::string& x) { return std::find(names.begin(), names.end(), x) != names.end(); };
            if (v == "resname") return residues([&](size_t r) { return in_names(topo.residue_name(r)); });
            const std::vector<std::string>& arr = v == "element" ? topo.elements : topo.names;
            Sel s; s.mask.resize(topo.n);
            for (size_t a = 0; a < topo.n; ++a) s.mask[a] = in_names(arr[a]);
            return s;
        }
        if (v == "resid") {
            // the residue sequence number of the file (PDB resSeq: may start anywhere and restart per chain), NOT the residue index:
            // VIAMD emits both forms side by side (src/main.cpp:2843-2848)
            take("(");
            const long a = integer();
            long b = a;
            if (accept(":")) b = integer();
            take(")");
            if (b < a) fail("bad range %ld:%ld", a, b);
            if (topo.res_seq.empty()) fail("resid(): the topology carries no residue sequence numbers (vmd_topology_t.residue_seq_id); use residue() for the 1-based residue index");
            Sel s = residues([&](size_t r) { return (long)topo.res_seq[r] >= a && (long)topo.res_seq[r] <= b; });
            if (s.structs.empty()) fail("resid(%ld:%ld) matches no residue", a, b);
            return s;
        }
        if (v == "residue" || v == "atom") {
            take("(");
            long a, b;
            range(a, b);
            take(")");
            if (v == "atom") return atom_range(a, b);
            if ((size_t)b > topo.nres) fail("%s(%ld) out of range (system has %zu residues)", v.c_str(), b, topo.nres);
            return residues([&](size_t r) { return (size_t)a <= r && r < (size_t)b; });
        }
        if (is_word("(")) fail("unsupported function '%s' (outside the rdf / sdf / distance path)", v.c_str());
        fail("unknown identifier '%s'", v.c_str());
    }
};

vmd_distance_kind_t dist_kind(const std::string& f) {
    if (f == "distance") return VMD_DISTANCE_COM;
    if (f == "distance_min") return VMD_DISTANCE_MIN;
    if (f == "distance_max") return VMD_DISTANCE_MAX;
    return VMD_DISTANCE_PAIR;
}

struct Skipped { std::string names; size_t beg, end; std::string reason; };
struct Report {
    std::vector<Skipped> skipped;
    std::vector<vmd_script_skipped_t> view;
    std::string fallback_source;
};

// One statement at a time.  A statement is parsed completely - up to its ';' - before its descriptor is appended, so a statement that
// fails leaves neither a property nor an identifier behind.  report == nullptr: the first error ends the compilation (the strict
// mode).  Otherwise the failing statement is recorded {left-hand names, source range, reason} and the next one is compiled: what a host
// hands to the evaluator it falls back on (include/vmd_md_script_shim.h), e.g. `a1 = angle(2,1,3) in resname("ALA");` and
// `{lin,plan,iso} = shape_weights(all);` of VIAMD's default script (src/main.cpp:528).  A later statement that uses an identifier of a
// skipped one is skipped with it ("unknown identifier").
void compile(vmd_script_ir_t* ir, const char* source, const vmd_topology_t* t, Report* report) {
    const Topo topo(t);
    const std::vector<Token> toks = tokenize(source, report != nullptr);
    std::map<std::string, Sel> env;
    Parser p(toks, topo, env);
    auto check = [](bool ok) { if (!ok) throw ScriptError(vmd_last_error()); };
    std::string fallback(source);
    // a compiled property that a SKIPPED statement uses (`d = distance(1,2); x = d * 2;`) must stay in the fallback's text, or
    // mdlib cannot compile the reduced script.  Blanked statements {left-hand name, token range, byte range} and the token ranges of the
    // skipped ones are recorded; after the pass every blanked statement whose name a kept statement mentions is restored, transitively
    struct Blanked { std::string name; size_t tok_first, tok_end, beg, end; bool kept; };
    std::vector<Blanked> blanked;
    std::vector<std::pair<size_t, size_t>> skipped_toks;
    while (p.peek().kind != T_END) {
        if (p.accept(";")) continue;
        const size_t first = p.i;
        // the statement's extent: up to the next ';' outside brackets (or the end of the text)
        size_t last = first;
        {
            int depth = 0;
            for (; last < toks.size(); ++last) {
                const Token& k = toks[last];
                if (k.kind != T_OP) continue;
                if (k.text == "(" || k.text == "[" || k.text == "{") depth += 1;
                else if (k.text == ")" || k.text == "]" || k.text == "}") depth -= 1;
                else if (k.text == ";" && depth <= 0) break;
            }
        }
        std::string names;
        std::function<void()> commit;
        bool is_property = false;
        try {
            if (p.is_word("{")) {
                // tuple assignment `{a, b, c} = f(...)`: no hot-path function returns a tuple
                p.take("{");
                names = p.take(nullptr, T_ID);
                while (p.accept(",")) names += "," + p.take(nullptr, T_ID);
                p.take("}");
                p.take("=");
                const Token& f = p.peek();
                fail("unsupported %s '%s' (outside the rdf / sdf / distance path)", f.kind == T_ID ? "function" : "expression", f.text.c_str());
            }
            const std::string name = p.take(nullptr, T_ID);
            names = name;
            p.take("=");
            const Token k = p.peek();
            const bool is_func = k.kind == T_ID && (k.text == "rdf" || k.text == "sdf" || k.text == "distance" || k.text == "distance_min" ||
                                                    k.text == "distance_max" || k.text == "distance_pair");
            if (is_func) {
                is_property = true;
                const std::string v = k.text;
                ++p.i;
                p.take("(");
                if (v == "rdf") {
                    const Sel ref = p.sel_or(); p.take(",");
                    const Sel tgt = p.sel_or(); p.take(",");
                    double rmin = 0.0, rmax;
                    if (p.accept("{")) { rmin = p.number(); p.take(","); rmax = p.number(); p.take("}"); }
                    else {
                        rmax = p.number();
                        if (p.accept(":")) { rmin = rmax; rmax = p.number(); }
                    }
                    p.take(")");
                    const auto a = ref.indices(), b = tgt.indices();
                    if (a.empty() || b.empty()) fail("%s: empty selection", name.c_str());
                    commit = [=]() { if (!vmd_ir_add_rdf(ir, name.c_str(), a.data(), a.size(), b.data(), b.size(), (float)rmin, (float)rmax)) throw ScriptError(vmd_last_error()); };
                } else if (v == "sdf") {
                    const Sel ref = p.sel_or(); p.take(",");
                    const Sel tgt = p.sel_or(); p.take(",");
                    const double cutoff = p.number();
                    p.take(")");
                    std::vector<std::vector<int32_t>> structs = ref.has_structs ? ref.structs : std::vector<std::vector<int32_t>>{ref.indices()};
                    if (structs.empty()) fail("%s: sdf reference structures must be non-empty and of equal size", name.c_str());
                    const size_t m = structs[0].size();
                    std::vector<int32_t> flat;
                    for (auto& st : structs) {
                        if (st.size() != m || m == 0) fail("%s: sdf reference structures must be non-empty and of equal size", name.c_str());
                        flat.insert(flat.end(), st.begin(), st.end());
                    }
                    const auto tg = tgt.indices();
                    const size_t K = structs.size();
                    commit = [=]() { if (!vmd_ir_add_sdf(ir, name.c_str(), flat.data(), K, m, tg.data(), tg.size(), (float)cutoff)) throw ScriptError(vmd_last_error()); };
                } else {
                    // the arguments may be followed by `in <contexts>`: find the closing parenthesis first
                    const size_t start = p.i;
                    size_t j = p.i;
                    int depth = 1;
                    while (depth) {
                        if (j >= toks.size()) fail("%s: missing ')'", name.c_str());
                        if (toks[j].kind == T_OP && toks[j].text == "(") depth += 1;
                        if (toks[j].kind == T_OP && toks[j].text == ")") depth -= 1;
                        ++j;
                    }
                    if (j < toks.size() && toks[j].kind == T_ID && toks[j].text == "in") {
                        Parser q(toks, topo, env);
                        q.i = j + 1;
                        const Sel ctx = q.sel_or();
                        if (!ctx.has_structs || ctx.structs.empty()) fail("%s: `in` needs an array of structures (residue(...), resname(...))", name.c_str());
                        std::vector<int32_t> a_all, b_all, a_off{0}, b_off{0};
                        for (auto& st : ctx.structs) {
                            Parser r(toks, topo, env, &st);
                            r.i = start;
                            const Sel a = r.sel_or(); r.take(","); const Sel b = r.sel_or(); r.take(")");
                            const auto ai = a.indices(), bi = b.indices();
                            if (ai.empty() || bi.empty()) fail("%s: empty selection inside a context", name.c_str());
                            a_all.insert(a_all.end(), ai.begin(), ai.end()); a_off.push_back((int32_t)a_all.size());
                            b_all.insert(b_all.end(), bi.begin(), bi.end()); b_off.push_back((int32_t)b_all.size());
                        }
                        p.i = q.i;
                        const size_t P = ctx.structs.size();
                        const vmd_distance_kind_t kind = dist_kind(v);
                        commit = [=]() {
                            if (!vmd_ir_add_distance_population(ir, name.c_str(), kind, P, a_all.data(), a_off.data(), b_all.data(), b_off.data())) throw ScriptError(vmd_last_error());
                        };
                    } else {
                        const Sel a = p.sel_or(); p.take(",");
                        const Sel b = p.sel_or();
                        p.take(")");
                        const auto ai = a.indices(), bi = b.indices();
                        const vmd_distance_kind_t kind = dist_kind(v);
                        commit = [=]() { if (!vmd_ir_add_distance(ir, name.c_str(), kind, ai.data(), ai.size(), bi.data(), bi.size())) throw ScriptError(vmd_last_error()); };
                    }
                }
            } else {
                const Sel sel = p.sel_or();
                commit = [&env, name, sel]() { env[name] = sel; };
            }
            if (p.peek().kind != T_END && !p.is_word(";")) fail("expected ;, found '%s'", p.peek().text.c_str());
            commit();
            if (p.peek().kind != T_END) p.take(";");
            if (is_property && report) {
                // the fallback evaluates the text WITHOUT this statement: blanked in place, so that every other offset stays what the editor shows
                const size_t e = last < toks.size() ? toks[last].end : toks[last - 1].end;
                for (size_t c = toks[first].beg; c < e && c < fallback.size(); ++c) if (fallback[c] != '\n') fallback[c] = ' ';
                blanked.push_back(Blanked{names, first, last, toks[first].beg, std::min(e, fallback.size()), false});
            }
        } catch (const ScriptError& e) {
            if (!report) throw;
            (void)check;
            if (names.empty()) names = toks[first].text;
            const size_t send = last > first ? toks[last - 1].end : toks[first].end;
            report->skipped.push_back({names, toks[first].beg, send, e.what()});
            skipped_toks.emplace_back(first, last);
            p.i = last < toks.size() ? last + 1 : last;
        }
    }
    if (report) {
        std::set<std::string> used;
        // identifiers on the right-hand side of a statement (everything after its first `=`; the whole statement when it has none)
        auto collect = [&](size_t a, size_t b) {
            size_t k = a;
            for (size_t q = a; q < b && q < toks.size(); ++q) if (toks[q].kind == T_OP && toks[q].text == "=") { k = q + 1; break; }
            for (; k < b && k < toks.size(); ++k) if (toks[k].kind == T_ID) used.insert(toks[k].text);
        };
        for (const auto& r : skipped_toks) collect(r.first, r.second);
        for (bool changed = true; changed;) {
            changed = false;
            for (Blanked& b : blanked) {
                if (b.kept || !used.count(b.name)) continue;
                b.kept = changed = true;
                for (size_t c = b.beg; c < b.end; ++c) fallback[c] = source[c];        // the GPU still evaluates it; mdlib evaluates it too, for its users
                collect(b.tok_first, b.tok_end);
            }
        }
        report->fallback_source = fallback;
    }
}

}  // namespace

extern "C" bool vmd_ir_compile_from_source(vmd_script_ir_t* ir, const char* source, const vmd_topology_t* topology) {
    if (!ir || !source || !topology) { vmd_set_last_error("vmd_ir_compile_from_source: NULL argument"); return false; }
    try {
        compile(ir, source, topology, nullptr);
    } catch (const std::exception& e) {
        vmd_set_last_error(e.what());
        return false;
    }
    return true;
}

struct vmd_script_report_t { Report r; };

extern "C" bool vmd_ir_compile_from_source_partial(vmd_script_ir_t* ir, const char* source, const vmd_topology_t* topology, vmd_script_report_t** report) {
    if (report) *report = nullptr;
    if (!ir || !source || !topology || !report) { vmd_set_last_error("vmd_ir_compile_from_source_partial: NULL argument"); return false; }
    std::unique_ptr<vmd_script_report_t> rep(new vmd_script_report_t());
    try {
        compile(ir, source, topology, &rep->r);
    } catch (const std::exception& e) {          // the topology itself is malformed: nothing a fallback could take over
        vmd_set_last_error(e.what());
        return false;
    }
    for (const Skipped& k : rep->r.skipped) rep->r.view.push_back(vmd_script_skipped_t{k.names.c_str(), k.beg, k.end, k.reason.c_str()});
    *report = rep.release();
    return true;
}
extern "C" size_t vmd_script_report_skipped_count(const vmd_script_report_t* r) { return r ? r->r.view.size() : 0; }
extern "C" const vmd_script_skipped_t* vmd_script_report_skipped(const vmd_script_report_t* r) { return r && !r->r.view.empty() ? r->r.view.data() : nullptr; }
extern "C" const char* vmd_script_report_fallback_source(const vmd_script_report_t* r) { return r ? r->r.fallback_source.c_str() : ""; }
extern "C" void vmd_script_report_free(vmd_script_report_t* r) { delete r; }

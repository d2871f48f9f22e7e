// integration/mi355x/framework/anakin_bin_model.h - the `.anakin.bin` model file WITHOUT a protobuf library: a reader and a writer of the
// protobuf wire format for the four schemas the reference's model files use (framework/model_parser/proto/graph.proto, node.proto,
// tensor.proto, operator.proto - field numbers and types below are those files'), into / from plain structs.
//
// Why: the reference's model parser (framework/model_parser/parser/parser.cpp:29-125, model_io.cpp) consumes protoc-generated classes and
// libprotobuf (or nanopb's runtime); neither exists in this image, so through round 5 Graph::load on this target's build read a TEXT model
// only and "the same .anakin.bin model" of the north star was never opened. The wire format itself is small: varints, 4 / 8-byte
// little-endian scalars, length-prefixed bytes; a message is a sequence of (field number << 3 | wire type) keys. This header has no
// dependency on the reference's headers (plain C++11): `tests/cpp_host/anakin_bin_tool.cpp` compiles it alone, and
// `tests/test_anakin_bin.py` checks it both ways against the OFFICIAL protobuf runtime (python `google.protobuf`, descriptors built from the
// schema) - files written here parse there field for field, files serialised there load here.
// proto3 rules honoured: repeated numeric scalars are written packed and accepted packed or one by one; scalar fields at their default are
// not written (members of valueType's `oneof data` are, when selected); unknown fields are skipped; a map is a repeated {1: key, 2: value}
// entry; int32 / enum negatives travel as 10-byte varints.
// Reference-side glue of the MI355X target's build (integration/), not part of the product library.
#ifndef ANAKIN_MI355X_ANAKIN_BIN_MODEL_H
#define ANAKIN_MI355X_ANAKIN_BIN_MODEL_H
#include <cstdint>
#include <cstring>
#include <map>
#include <string>
#include <utility>
#include <vector>

namespace anakin_bin {

// tensor.proto: DateTypeProto
enum DType { DT_STR = 0, DT_INT8 = 2, DT_INT32 = 4, DT_FLOAT16 = 8, DT_FLOAT = 13, DT_DOUBLE = 14, DT_BOOLEN = 20, DT_CACHE_LIST = 30, DT_TENSOR = 31 };

struct Cache {                       // tensor.proto: CacheDate
    std::vector<std::string> s;      // 1
    std::vector<int32_t> i;          // 2
    std::vector<float> f;            // 3
    std::vector<uint8_t> b;          // 4 (bool)
    std::vector<Cache> l;            // 5
    int type = 0;                    // 6
    int64_t size = 0;                // 7
    std::string c;                   // 8 (int8 payload)
};
struct Dim {                         // tensor.proto: TensorShape.Dim, as TensorShape.dim (3)
    std::vector<int32_t> value;      // 1
    int64_t size = 0;                // 2
    bool present = false;            // the TensorShape message was on the wire
};
struct Tensor {                      // tensor.proto: TensorProto
    std::string name;                // 1
    bool shared = false;             // 2
    std::string share_from;          // 3
    Dim shape, valid_shape;          // 8, 9
    Cache data, scale;               // 10, 11
};
struct Value {                       // node.proto: valueType (`oneof data` + type)
    std::string s;                   // 1
    int32_t i = 0;                   // 2
    float f = 0.f;                   // 3
    bool b = false;                  // 4
    Cache cache_list;                // 8
    Tensor tensor;                   // 10
    int type = 0;                    // 14
};
struct Op {                          // operator.proto: OpProto
    std::string name, description;   // 1, 5
    bool is_commutative = false;     // 2
    int32_t in_num = 0, out_num = 0; // 3, 4
    bool present = false;            // the message was on the wire (a writer that fills a node's Op sets it)
};
struct Node {                        // node.proto: NodeProto
    std::string name;                // 1
    std::vector<std::string> ins, outs;                    // 2, 3
    std::vector<std::pair<std::string, Value> > attr;      // 10 (map entries in file order; a repeated key: the last one counts)
    int32_t lane = 0;                // 11
    bool need_wait = false;          // 12
    Op op;                           // 15
    int bit_type = 0;                // 16
};
struct Target {                      // graph.proto: TargetProto
    std::string node;                // 1
    std::vector<float> scale;        // 2
    int layout = 0;                  // 3
};
struct List {                        // graph.proto: List
    std::vector<std::string> val;    // 1
    std::vector<Target> target;      // 2
};
struct Graph {                       // graph.proto: GraphProto
    std::string name;                // 1
    std::vector<Node> nodes;         // 2
    std::vector<std::pair<std::string, List> > edges_in, edges_out;       // 3, 4 (file order)
    std::map<std::string, Tensor> edges_info;                             // 5
    std::vector<std::string> ins, outs;                                   // 6, 7
    int32_t ver_major = 0, ver_minor = 0, ver_patch = 0;                  // 10: Version 1, 2, 3
    int64_t ver_version = 0;                                              //     4
    bool has_version = false;
    int32_t temp_mem_used = 0, original_temp_mem_used = 0, system_mem_used = 0, model_mem_used = 0;      // 11: Info 1 .. 4
    bool is_optimized = false;                                            //     10
    bool has_summary = false;
};

// ---------------------------------------------------------------------------------------------------------------------------------
class Reader {
public:
    Reader(const uint8_t* p, size_t n) : p_(p), end_(p + n) {}
    bool ok() const { return ok_; }
    bool done() const { return p_ >= end_ || !ok_; }
    uint64_t varint() {
        uint64_t v = 0;
        for (int shift = 0; shift < 70; shift += 7) {
            if (p_ >= end_) return fail();
            const uint8_t b = *p_++;
            if (shift < 64) v |= (uint64_t)(b & 0x7f) << shift;
            if (!(b & 0x80)) return v;
        }
        return fail();
    }
    bool key(int* field, int* wt) {
        const uint64_t k = varint();
        *field = (int)(k >> 3);
        *wt = (int)(k & 7);
        if (ok_ && *field <= 0) fail();      // field number 0 does not exist: a corrupted key, not the end of the message
        return ok_;
    }
    uint32_t fixed32() {
        if (end_ - p_ < 4) return (uint32_t)fail();
        uint32_t v;
        std::memcpy(&v, p_, 4);
        p_ += 4;
        return v;
    }
    float f32() {
        const uint32_t u = fixed32();
        float f;
        std::memcpy(&f, &u, 4);
        return f;
    }
    Reader sub() {                   // a length-delimited field's payload
        const uint64_t n = varint();
        if (!ok_ || n > (uint64_t)(end_ - p_)) { fail(); return Reader(p_, 0); }
        Reader r(p_, (size_t)n);
        p_ += n;
        return r;
    }
    std::string str() {
        Reader r = sub();
        return std::string((const char*)r.p_, (size_t)(r.end_ - r.p_));
    }
    void skip(int wt) {
        switch (wt) {
        case 0: (void)varint(); break;
        case 1: if (end_ - p_ < 8) fail(); else p_ += 8; break;
        case 2: (void)sub(); break;
        case 5: (void)fixed32(); break;
        default: fail();             // groups (3 / 4) do not occur in proto3 files
        }
    }
    // repeated numeric scalar: one element (wire type 0 / 5) or a packed run (wire type 2)
    void ints(int wt, std::vector<int32_t>& out) {
        if (wt == 2) { Reader r = sub(); while (!r.done()) out.push_back((int32_t)r.varint()); if (!r.ok()) fail(); }
        else if (wt == 0) out.push_back((int32_t)varint());
        else skip(wt);               // (a wire type the field cannot have: an unknown field, as in every protobuf parser)
    }
    void bools(int wt, std::vector<uint8_t>& out) {
        if (wt == 2) { Reader r = sub(); while (!r.done()) out.push_back(r.varint() != 0); if (!r.ok()) fail(); }
        else if (wt == 0) out.push_back(varint() != 0);
        else skip(wt);
    }
    void floats(int wt, std::vector<float>& out) {
        if (wt == 2) {
            Reader r = sub();
            const size_t n = (size_t)(r.end_ - r.p_);
            if (n % 4) { fail(); return; }
            const size_t at = out.size();
            out.resize(at + n / 4);
            if (n) std::memcpy(out.data() + at, r.p_, n);     // (little-endian host: the wire order)
        } else if (wt == 5) out.push_back(f32());
        else skip(wt);
    }
    void mark_bad() { fail(); }
private:
    uint64_t fail() { ok_ = false; p_ = end_; return 0; }
    const uint8_t* p_;
    const uint8_t* end_;
    bool ok_ = true;
};

inline void read_cache(Reader r, Cache& c, Reader& parent, int depth = 0) {
    int f, wt;
    while (!r.done() && r.key(&f, &wt)) {
        switch (f) {
        case 1: if (wt == 2) c.s.push_back(r.str()); else r.skip(wt); break;
        case 2: r.ints(wt, c.i); break;
        case 3: r.floats(wt, c.f); break;
        case 4: r.bools(wt, c.b); break;
        case 5:
            if (wt != 2) r.skip(wt);
            else if (depth < 8) { c.l.emplace_back(); read_cache(r.sub(), c.l.back(), r, depth + 1); }
            else r.mark_bad();       // lists nested deeper than any model holds (the format has lists of lists): refused, not recursed into
            break;
        case 6: if (wt == 0) c.type = (int)(int32_t)r.varint(); else r.skip(wt); break;
        case 7: if (wt == 0) c.size = (int64_t)r.varint(); else r.skip(wt); break;
        case 8: if (wt == 2) c.c = r.str(); else r.skip(wt); break;
        default: r.skip(wt);
        }
    }
    if (!r.ok()) parent.mark_bad();
}
inline void read_shape(Reader r, Dim& d, Reader& parent) {      // TensorShape { Dim dim = 3 }
    d.present = true;
    int f, wt;
    while (!r.done() && r.key(&f, &wt)) {
        if (f == 3 && wt == 2) {
            Reader q = r.sub();
            int g, wu;
            while (!q.done() && q.key(&g, &wu)) {
                if (g == 1) q.ints(wu, d.value);
                else if (g == 2 && wu == 0) d.size = (int64_t)q.varint();
                else q.skip(wu);
            }
            if (!q.ok()) r.mark_bad();
        } else r.skip(wt);
    }
    if (!r.ok()) parent.mark_bad();
}
inline void read_tensor(Reader r, Tensor& t, Reader& parent) {
    int f, wt;
    while (!r.done() && r.key(&f, &wt)) {
        switch (f) {
        case 1: if (wt == 2) t.name = r.str(); else r.skip(wt); break;
        case 2: if (wt == 0) t.shared = r.varint() != 0; else r.skip(wt); break;
        case 3: if (wt == 2) t.share_from = r.str(); else r.skip(wt); break;
        case 8: if (wt == 2) read_shape(r.sub(), t.shape, r); else r.skip(wt); break;
        case 9: if (wt == 2) read_shape(r.sub(), t.valid_shape, r); else r.skip(wt); break;
        case 10: if (wt == 2) read_cache(r.sub(), t.data, r); else r.skip(wt); break;
        case 11: if (wt == 2) read_cache(r.sub(), t.scale, r); else r.skip(wt); break;
        default: r.skip(wt);
        }
    }
    if (!r.ok()) parent.mark_bad();
}
inline void read_value(Reader r, Value& v, Reader& parent) {
    int f, wt;
    while (!r.done() && r.key(&f, &wt)) {
        switch (f) {
        case 1: if (wt == 2) v.s = r.str(); else r.skip(wt); break;
        case 2: if (wt == 0) v.i = (int32_t)r.varint(); else r.skip(wt); break;
        case 3: if (wt == 5) v.f = r.f32(); else r.skip(wt); break;
        case 4: if (wt == 0) v.b = r.varint() != 0; else r.skip(wt); break;
        case 8: if (wt == 2) read_cache(r.sub(), v.cache_list, r); else r.skip(wt); break;
        case 10: if (wt == 2) read_tensor(r.sub(), v.tensor, r); else r.skip(wt); break;
        case 14: if (wt == 0) v.type = (int)(int32_t)r.varint(); else r.skip(wt); break;
        default: r.skip(wt);
        }
    }
    if (!r.ok()) parent.mark_bad();
}
inline void read_node(Reader r, Node& n, Reader& parent) {
    int f, wt;
    while (!r.done() && r.key(&f, &wt)) {
        switch (f) {
        case 1: if (wt == 2) n.name = r.str(); else r.skip(wt); break;
        case 2: if (wt == 2) n.ins.push_back(r.str()); else r.skip(wt); break;
        case 3: if (wt == 2) n.outs.push_back(r.str()); else r.skip(wt); break;
        case 10:
            if (wt == 2) {
                Reader e = r.sub();
                n.attr.emplace_back();
                int g, wu;
                while (!e.done() && e.key(&g, &wu)) {
                    if (g == 1 && wu == 2) n.attr.back().first = e.str();
                    else if (g == 2 && wu == 2) read_value(e.sub(), n.attr.back().second, e);
                    else e.skip(wu);
                }
                if (!e.ok()) r.mark_bad();
            } else r.skip(wt);
            break;
        case 11: if (wt == 0) n.lane = (int32_t)r.varint(); else r.skip(wt); break;
        case 12: if (wt == 0) n.need_wait = r.varint() != 0; else r.skip(wt); break;
        case 15:
            if (wt == 2) {
                Reader o = r.sub();
                n.op.present = true;
                int g, wu;
                while (!o.done() && o.key(&g, &wu)) {
                    if (g == 1 && wu == 2) n.op.name = o.str();
                    else if (g == 2 && wu == 0) n.op.is_commutative = o.varint() != 0;
                    else if (g == 3 && wu == 0) n.op.in_num = (int32_t)o.varint();
                    else if (g == 4 && wu == 0) n.op.out_num = (int32_t)o.varint();
                    else if (g == 5 && wu == 2) n.op.description = o.str();
                    else o.skip(wu);
                }
                if (!o.ok()) r.mark_bad();
            } else r.skip(wt);
            break;
        case 16: if (wt == 0) n.bit_type = (int)(int32_t)r.varint(); else r.skip(wt); break;
        default: r.skip(wt);
        }
    }
    if (!r.ok()) parent.mark_bad();
}
inline void read_list(Reader r, List& l, Reader& parent) {
    int f, wt;
    while (!r.done() && r.key(&f, &wt)) {
        if (f == 1 && wt == 2) l.val.push_back(r.str());
        else if (f == 2 && wt == 2) {
            Reader t = r.sub();
            l.target.emplace_back();
            Target& tg = l.target.back();
            int g, wu;
            while (!t.done() && t.key(&g, &wu)) {
                if (g == 1 && wu == 2) tg.node = t.str();
                else if (g == 2) t.floats(wu, tg.scale);
                else if (g == 3 && wu == 0) tg.layout = (int)(int32_t)t.varint();
                else t.skip(wu);
            }
            if (!t.ok()) r.mark_bad();
        } else r.skip(wt);
    }
    if (!r.ok()) parent.mark_bad();
}

// false: not a well-formed GraphProto (truncated, a length running past its parent, an impossible wire type)
inline bool decode(const uint8_t* data, size_t len, Graph& g) {
    Reader r(data, len);
    int f, wt;
    while (!r.done() && r.key(&f, &wt)) {
        switch (f) {
        case 1: if (wt == 2) g.name = r.str(); else r.skip(wt); break;
        case 2: if (wt == 2) { g.nodes.emplace_back(); read_node(r.sub(), g.nodes.back(), r); } else r.skip(wt); break;
        case 3: case 4:
            if (wt == 2) {
                Reader e = r.sub();
                auto& vec = f == 3 ? g.edges_in : g.edges_out;
                vec.emplace_back();
                int k, wu;
                while (!e.done() && e.key(&k, &wu)) {
                    if (k == 1 && wu == 2) vec.back().first = e.str();
                    else if (k == 2 && wu == 2) read_list(e.sub(), vec.back().second, e);
                    else e.skip(wu);
                }
                if (!e.ok()) r.mark_bad();
            } else r.skip(wt);
            break;
        case 5:
            if (wt == 2) {
                Reader e = r.sub();
                std::string key;
                Tensor t;
                int k, wu;
                while (!e.done() && e.key(&k, &wu)) {
                    if (k == 1 && wu == 2) key = e.str();
                    else if (k == 2 && wu == 2) read_tensor(e.sub(), t, e);
                    else e.skip(wu);
                }
                if (!e.ok()) r.mark_bad();
                g.edges_info[key] = t;
            } else r.skip(wt);
            break;
        case 6: if (wt == 2) g.ins.push_back(r.str()); else r.skip(wt); break;
        case 7: if (wt == 2) g.outs.push_back(r.str()); else r.skip(wt); break;
        case 10:
            if (wt == 2) {
                Reader v = r.sub();
                g.has_version = true;
                int k, wu;
                while (!v.done() && v.key(&k, &wu)) {
                    if (wu != 0) { v.skip(wu); continue; }
                    const uint64_t x = v.varint();
                    if (k == 1) g.ver_major = (int32_t)x; else if (k == 2) g.ver_minor = (int32_t)x;
                    else if (k == 3) g.ver_patch = (int32_t)x; else if (k == 4) g.ver_version = (int64_t)x;
                }
                if (!v.ok()) r.mark_bad();
            } else r.skip(wt);
            break;
        case 11:
            if (wt == 2) {
                Reader v = r.sub();
                g.has_summary = true;
                int k, wu;
                while (!v.done() && v.key(&k, &wu)) {
                    if (wu != 0) { v.skip(wu); continue; }
                    const uint64_t x = v.varint();
                    if (k == 1) g.temp_mem_used = (int32_t)x; else if (k == 2) g.original_temp_mem_used = (int32_t)x;
                    else if (k == 3) g.system_mem_used = (int32_t)x; else if (k == 4) g.model_mem_used = (int32_t)x;
                    else if (k == 10) g.is_optimized = x != 0;
                }
                if (!v.ok()) r.mark_bad();
            } else r.skip(wt);
            break;
        default: r.skip(wt);
        }
    }
    return r.ok();
}

// ---------------------------------------------------------------------------------------------------------------------------------
class Writer {
public:
    std::string buf;
    void varint(uint64_t v) {
        while (v >= 0x80) { buf.push_back((char)(v | 0x80)); v >>= 7; }
        buf.push_back((char)v);
    }
    void key(int field, int wt) { varint(((uint64_t)field << 3) | (uint64_t)wt); }
    void i32(int field, int32_t v, bool always = false) { if (v || always) { key(field, 0); varint((uint64_t)(int64_t)v); } }
    void i64(int field, int64_t v) { if (v) { key(field, 0); varint((uint64_t)v); } }
    void boolean(int field, bool v, bool always = false) { if (v || always) { key(field, 0); varint(v ? 1 : 0); } }
    void f32(int field, float v, bool always = false) {
        uint32_t u;
        std::memcpy(&u, &v, 4);
        if (u || always) { key(field, 5); buf.append((const char*)&u, 4); }
    }
    void bytes(int field, const std::string& s, bool always = false) { if (!s.empty() || always) { key(field, 2); varint(s.size()); buf += s; } }
    void message(int field, const Writer& m) { key(field, 2); varint(m.buf.size()); buf += m.buf; }
    void packed_i32(int field, const std::vector<int32_t>& v) {
        if (v.empty()) return;
        Writer p;
        for (int32_t x : v) p.varint((uint64_t)(int64_t)x);
        message(field, p);
    }
    void packed_bool(int field, const std::vector<uint8_t>& v) {
        if (v.empty()) return;
        key(field, 2); varint(v.size());
        for (uint8_t x : v) buf.push_back(x ? 1 : 0);
    }
    void packed_f32(int field, const std::vector<float>& v) {
        if (v.empty()) return;
        key(field, 2); varint(v.size() * 4);
        buf.append((const char*)v.data(), v.size() * 4);
    }
};

inline void write_cache(Writer& w, const Cache& c) {
    for (auto& s : c.s) w.bytes(1, s, true);
    w.packed_i32(2, c.i);
    w.packed_f32(3, c.f);
    w.packed_bool(4, c.b);
    for (auto& l : c.l) { Writer m; write_cache(m, l); w.message(5, m); }
    w.i32(6, c.type);
    w.i64(7, c.size);
    w.bytes(8, c.c);
}
inline bool cache_empty(const Cache& c) { return c.s.empty() && c.i.empty() && c.f.empty() && c.b.empty() && c.l.empty() && !c.type && !c.size && c.c.empty(); }
inline void write_shape(Writer& w, int field, const Dim& d) {
    if (!d.present && d.value.empty() && !d.size) return;
    Writer dim, sh;
    dim.packed_i32(1, d.value);
    dim.i64(2, d.size);
    sh.message(3, dim);
    w.message(field, sh);
}
inline void write_tensor(Writer& w, const Tensor& t) {
    w.bytes(1, t.name);
    w.boolean(2, t.shared);
    w.bytes(3, t.share_from);
    write_shape(w, 8, t.shape);
    write_shape(w, 9, t.valid_shape);
    if (!cache_empty(t.data)) { Writer m; write_cache(m, t.data); w.message(10, m); }
    if (!cache_empty(t.scale)) { Writer m; write_cache(m, t.scale); w.message(11, m); }
}
inline void write_value(Writer& w, const Value& v) {
    switch (v.type) {                 // the selected member of `oneof data` is on the wire even at its default
    case DT_STR: w.bytes(1, v.s, true); break;
    case DT_INT32: w.i32(2, v.i, true); break;
    case DT_FLOAT: case DT_DOUBLE: w.f32(3, v.f, true); break;
    case DT_BOOLEN: w.boolean(4, v.b, true); break;
    case DT_CACHE_LIST: { Writer m; write_cache(m, v.cache_list); w.message(8, m); } break;
    case DT_TENSOR: { Writer m; write_tensor(m, v.tensor); w.message(10, m); } break;
    default: break;
    }
    w.i32(14, v.type);
}
inline void write_node(Writer& w, const Node& n) {
    w.bytes(1, n.name);
    for (auto& s : n.ins) w.bytes(2, s, true);
    for (auto& s : n.outs) w.bytes(3, s, true);
    for (auto& kv : n.attr) {
        Writer e, v;
        e.bytes(1, kv.first);
        write_value(v, kv.second);
        e.message(2, v);
        w.message(10, e);
    }
    w.i32(11, n.lane);
    w.boolean(12, n.need_wait);
    if (n.op.present || !n.op.name.empty() || !n.op.description.empty() || n.op.is_commutative || n.op.in_num || n.op.out_num) {
        Writer o;
        o.bytes(1, n.op.name);
        o.boolean(2, n.op.is_commutative);
        o.i32(3, n.op.in_num);
        o.i32(4, n.op.out_num);
        o.bytes(5, n.op.description);
        w.message(15, o);
    }
    w.i32(16, n.bit_type);
}
inline void write_list(Writer& w, const List& l) {
    for (auto& s : l.val) w.bytes(1, s, true);
    for (auto& t : l.target) {
        Writer m;
        m.bytes(1, t.node);
        m.packed_f32(2, t.scale);
        m.i32(3, t.layout);
        w.message(2, m);
    }
}
inline std::string encode(const Graph& g) {
    Writer w;
    w.bytes(1, g.name);
    for (auto& n : g.nodes) { Writer m; write_node(m, n); w.message(2, m); }
    for (int f = 3; f <= 4; ++f)
        for (auto& kv : (f == 3 ? g.edges_in : g.edges_out)) {
            Writer e, l;
            e.bytes(1, kv.first);
            write_list(l, kv.second);
            e.message(2, l);
            w.message(f, e);
        }
    for (auto& kv : g.edges_info) {
        Writer e, t;
        e.bytes(1, kv.first);
        write_tensor(t, kv.second);
        e.message(2, t);
        w.message(5, e);
    }
    for (auto& s : g.ins) w.bytes(6, s, true);
    for (auto& s : g.outs) w.bytes(7, s, true);
    if (g.has_version) {
        Writer v;
        v.i32(1, g.ver_major); v.i32(2, g.ver_minor); v.i32(3, g.ver_patch); v.i64(4, g.ver_version);
        w.message(10, v);
    }
    if (g.has_summary || g.temp_mem_used || g.original_temp_mem_used || g.system_mem_used || g.model_mem_used || g.is_optimized) {
        Writer v;
        v.i32(1, g.temp_mem_used); v.i32(2, g.original_temp_mem_used); v.i32(3, g.system_mem_used); v.i32(4, g.model_mem_used);
        v.boolean(10, g.is_optimized);
        w.message(11, v);
    }
    return w.buf;
}

}  // namespace anakin_bin
#endif

// Protobuf wire-format decoding of the ONNX messages listed in onnx_proto.h.
// Field numbers follow the published onnx.proto3 schema (ONNX IR v3..v10).
#include "onnx_proto.h"

#include <cstdio>
#include <cstring>
#include <set>

namespace bn {
namespace {

struct Reader {
    const uint8_t *p, *end;
    Reader(const uint8_t *b, size_t n) : p(b), end(b + n) {}
    bool done() const { return p >= end; }
    uint64_t varint() {
        uint64_t v = 0;
        int shift = 0;
        while (true) {
            if (p >= end) throw OnnxParseError("truncated varint");
            uint8_t b = *p++;
            v |= (uint64_t)(b & 0x7f) << shift;
            if (!(b & 0x80)) break;
            shift += 7;
            if (shift > 63) throw OnnxParseError("varint too long");
        }
        return v;
    }
    // returns (field, wiretype)
    void key(uint32_t &field, uint32_t &wt) {
        uint64_t k = varint();
        field = (uint32_t)(k >> 3);
        wt = (uint32_t)(k & 7);
    }
    Reader sub() {
        uint64_t n = varint();
        if ((uint64_t)(end - p) < n) throw OnnxParseError("truncated length-delimited field");
        Reader r(p, (size_t)n);
        p += n;
        return r;
    }
    std::string str() {
        Reader r = sub();
        return std::string((const char *)r.p, (size_t)(r.end - r.p));
    }
    uint32_t fixed32() {
        if (end - p < 4) throw OnnxParseError("truncated fixed32");
        uint32_t v;
        memcpy(&v, p, 4);
        p += 4;
        return v;
    }
    uint64_t fixed64() {
        if (end - p < 8) throw OnnxParseError("truncated fixed64");
        uint64_t v;
        memcpy(&v, p, 8);
        p += 8;
        return v;
    }
    void skip(uint32_t wt) {
        switch (wt) {
            case 0: varint(); break;
            case 1: fixed64(); break;
            case 2: sub(); break;
            case 5: fixed32(); break;
            default: throw OnnxParseError("unsupported wire type " + std::to_string(wt));
        }
    }
};

float f32_from_bits(uint32_t b) {
    float f;
    memcpy(&f, &b, 4);
    return f;
}

float half_to_float(uint16_t h) {
    uint32_t sign = (uint32_t)(h & 0x8000) << 16;
    uint32_t exp = (h >> 10) & 0x1f, man = h & 0x3ff;
    uint32_t bits;
    if (exp == 0) {
        if (man == 0) bits = sign;
        else {
            int e = -1;
            do { man <<= 1; e++; } while (!(man & 0x400));
            bits = sign | ((uint32_t)(127 - 15 - e) << 23) | ((man & 0x3ff) << 13);
        }
    } else if (exp == 31) bits = sign | 0x7f800000u | (man << 13);
    else bits = sign | ((exp + 112) << 23) | (man << 13);
    return f32_from_bits(bits);
}

OnnxTensor parse_tensor(Reader r) {
    OnnxTensor t;
    std::string raw;
    bool has_raw = false;
    std::vector<float> fdata;
    std::vector<int64_t> i32data, i64data;
    std::vector<double> ddata;
    while (!r.done()) {
        uint32_t f, wt;
        r.key(f, wt);
        switch (f) {
            case 1:  // dims
                if (wt == 2) { Reader s = r.sub(); while (!s.done()) t.dims.push_back((int64_t)s.varint()); }
                else t.dims.push_back((int64_t)r.varint());
                break;
            case 2: t.data_type = (int32_t)r.varint(); break;
            case 4:  // float_data
                if (wt == 2) { Reader s = r.sub(); while (!s.done()) fdata.push_back(f32_from_bits(s.fixed32())); }
                else fdata.push_back(f32_from_bits(r.fixed32()));
                break;
            case 5:  // int32_data
                if (wt == 2) { Reader s = r.sub(); while (!s.done()) i32data.push_back((int64_t)(int32_t)s.varint()); }
                else i32data.push_back((int64_t)(int32_t)r.varint());
                break;
            case 7:  // int64_data
                if (wt == 2) { Reader s = r.sub(); while (!s.done()) i64data.push_back((int64_t)s.varint()); }
                else i64data.push_back((int64_t)r.varint());
                break;
            case 8: t.name = r.str(); break;
            case 9: raw = r.str(); has_raw = true; break;
            case 10:  // double_data
                if (wt == 2) { Reader s = r.sub(); while (!s.done()) { uint64_t b = s.fixed64(); double d; memcpy(&d, &b, 8); ddata.push_back(d); } }
                else { uint64_t b = r.fixed64(); double d; memcpy(&d, &b, 8); ddata.push_back(d); }
                break;
            case 13: case 14:
                if (f == 14) { if (r.varint() == 1) throw OnnxParseError("tensor '" + t.name + "' uses external data, which is not supported"); }
                else r.skip(wt);
                break;
            default: r.skip(wt);
        }
    }
    int64_t n = t.numel();
    auto need = [&](size_t have, size_t esz) {
        if (have != (size_t)n * esz)
            throw OnnxParseError("tensor '" + t.name + "': raw_data size " + std::to_string(have) +
                                 " does not match dims");
    };
    switch (t.data_type) {
        case 1:  // FLOAT
            if (has_raw) { need(raw.size(), 4); t.f.resize(n); memcpy(t.f.data(), raw.data(), raw.size()); }
            else t.f = fdata;
            break;
        case 11:  // DOUBLE
            if (has_raw) { need(raw.size(), 8); t.f.resize(n); for (int64_t k = 0; k < n; k++) { double d; memcpy(&d, raw.data() + 8 * k, 8); t.f[k] = (float)d; } }
            else { t.f.resize(ddata.size()); for (size_t k = 0; k < ddata.size(); k++) t.f[k] = (float)ddata[k]; }
            break;
        case 10:  // FLOAT16
            t.f.resize(n);
            if (has_raw) { need(raw.size(), 2); for (int64_t k = 0; k < n; k++) { uint16_t h; memcpy(&h, raw.data() + 2 * k, 2); t.f[k] = half_to_float(h); } }
            else for (int64_t k = 0; k < n && k < (int64_t)i32data.size(); k++) t.f[k] = half_to_float((uint16_t)i32data[k]);
            break;
        case 7:  // INT64
            if (has_raw) { need(raw.size(), 8); t.i.resize(n); memcpy(t.i.data(), raw.data(), raw.size()); }
            else t.i = i64data;
            break;
        case 6:  // INT32
            if (has_raw) { need(raw.size(), 4); t.i.resize(n); for (int64_t k = 0; k < n; k++) { int32_t v; memcpy(&v, raw.data() + 4 * k, 4); t.i[k] = v; } }
            else t.i = i32data;
            break;
        case 9: case 2: case 3:  // BOOL / UINT8 / INT8
            if (has_raw) { need(raw.size(), 1); t.i.resize(n); for (int64_t k = 0; k < n; k++) t.i[k] = t.data_type == 3 ? (int64_t)(int8_t)raw[k] : (int64_t)(uint8_t)raw[k]; }
            else t.i = i32data;
            break;
        default:
            throw OnnxParseError("tensor '" + t.name + "': unsupported data_type " + std::to_string(t.data_type));
    }
    size_t got = t.is_float() ? t.f.size() : t.i.size();
    if ((int64_t)got != n)
        throw OnnxParseError("tensor '" + t.name + "': " + std::to_string(got) + " values for " +
                             std::to_string(n) + " elements");
    return t;
}

OnnxAttr parse_attr(Reader r) {
    OnnxAttr a;
    while (!r.done()) {
        uint32_t f, wt;
        r.key(f, wt);
        switch (f) {
            case 1: a.name = r.str(); break;
            case 2: a.f = f32_from_bits(r.fixed32()); if (!a.type) a.type = 1; break;
            case 3: a.i = (int64_t)r.varint(); if (!a.type) a.type = 2; break;
            case 4: a.s = r.str(); if (!a.type) a.type = 3; break;
            case 5: a.t = parse_tensor(r.sub()); if (!a.type) a.type = 4; break;
            case 7:
                if (wt == 2) { Reader s = r.sub(); while (!s.done()) a.floats.push_back(f32_from_bits(s.fixed32())); }
                else a.floats.push_back(f32_from_bits(r.fixed32()));
                break;
            case 8:
                if (wt == 2) { Reader s = r.sub(); while (!s.done()) a.ints.push_back((int64_t)s.varint()); }
                else a.ints.push_back((int64_t)r.varint());
                break;
            case 20: a.type = (int32_t)r.varint(); break;
            default: r.skip(wt);
        }
    }
    return a;
}

OnnxNode parse_node(Reader r) {
    OnnxNode n;
    while (!r.done()) {
        uint32_t f, wt;
        r.key(f, wt);
        switch (f) {
            case 1: n.inputs.push_back(r.str()); break;
            case 2: n.outputs.push_back(r.str()); break;
            case 3: n.name = r.str(); break;
            case 4: n.op_type = r.str(); break;
            case 5: { OnnxAttr a = parse_attr(r.sub()); n.attrs[a.name] = std::move(a); break; }
            case 7: n.domain = r.str(); break;
            default: r.skip(wt);
        }
    }
    return n;
}

OnnxValueInfo parse_value_info(Reader r) {
    OnnxValueInfo v;
    while (!r.done()) {
        uint32_t f, wt;
        r.key(f, wt);
        if (f == 1) v.name = r.str();
        else if (f == 2) {  // TypeProto
            Reader tp = r.sub();
            while (!tp.done()) {
                uint32_t f2, wt2;
                tp.key(f2, wt2);
                if (f2 == 1) {  // tensor_type
                    Reader tt = tp.sub();
                    while (!tt.done()) {
                        uint32_t f3, wt3;
                        tt.key(f3, wt3);
                        if (f3 == 1) v.elem_type = (int32_t)tt.varint();
                        else if (f3 == 2) {  // TensorShapeProto
                            v.has_shape = true;
                            Reader sh = tt.sub();
                            while (!sh.done()) {
                                uint32_t f4, wt4;
                                sh.key(f4, wt4);
                                if (f4 == 1) {  // Dimension
                                    Reader d = sh.sub();
                                    int64_t val = -1;
                                    while (!d.done()) {
                                        uint32_t f5, wt5;
                                        d.key(f5, wt5);
                                        if (f5 == 1) val = (int64_t)d.varint();
                                        else d.skip(wt5);
                                    }
                                    v.shape.push_back(val);
                                } else sh.skip(wt4);
                            }
                        } else tt.skip(wt3);
                    }
                } else tp.skip(wt2);
            }
        } else r.skip(wt);
    }
    return v;
}

void parse_graph(Reader r, OnnxModel &m) {
    std::vector<OnnxValueInfo> inputs;
    while (!r.done()) {
        uint32_t f, wt;
        r.key(f, wt);
        switch (f) {
            case 1: m.nodes.push_back(parse_node(r.sub())); break;
            case 5: m.initializers.push_back(parse_tensor(r.sub())); break;
            case 11: inputs.push_back(parse_value_info(r.sub())); break;
            case 12: m.outputs.push_back(parse_value_info(r.sub())); break;
            case 15: throw OnnxParseError("sparse initializers are not supported");
            default: r.skip(wt);
        }
    }
    std::set<std::string> init_names;
    for (auto &t : m.initializers) init_names.insert(t.name);
    for (auto &v : inputs)
        if (!init_names.count(v.name)) m.inputs.push_back(v);
}

}  // namespace

OnnxModel parse_onnx(const uint8_t *data, size_t len) {
    OnnxModel m;
    Reader r(data, len);
    bool saw_graph = false;
    while (!r.done()) {
        uint32_t f, wt;
        r.key(f, wt);
        switch (f) {
            case 1: m.ir_version = (int64_t)r.varint(); break;
            case 2: m.producer = r.str(); break;
            case 7: parse_graph(r.sub(), m); saw_graph = true; break;
            case 8: {  // opset_import
                Reader o = r.sub();
                std::string domain;
                int64_t ver = 0;
                while (!o.done()) {
                    uint32_t f2, wt2;
                    o.key(f2, wt2);
                    if (f2 == 1) domain = o.str();
                    else if (f2 == 2) ver = (int64_t)o.varint();
                    else o.skip(wt2);
                }
                if (domain.empty() || domain == "ai.onnx") m.opset = ver;
                break;
            }
            default: r.skip(wt);
        }
    }
    if (!saw_graph) throw OnnxParseError("not an ONNX ModelProto: no graph field");
    return m;
}

OnnxModel parse_onnx_file(const std::string &path) {
    FILE *fp = fopen(path.c_str(), "rb");
    if (!fp) throw OnnxParseError("cannot open '" + path + "'");
    fseek(fp, 0, SEEK_END);
    long n = ftell(fp);
    fseek(fp, 0, SEEK_SET);
    std::vector<uint8_t> buf((size_t)(n > 0 ? n : 0));
    size_t got = buf.empty() ? 0 : fread(buf.data(), 1, buf.size(), fp);
    fclose(fp);
    if (got != buf.size()) throw OnnxParseError("short read on '" + path + "'");
    if (buf.empty()) throw OnnxParseError("'" + path + "' is empty");
    return parse_onnx(buf.data(), buf.size());
}

}  // namespace bn

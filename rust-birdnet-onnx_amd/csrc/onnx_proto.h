// Minimal ONNX (protobuf wire format) reader: just enough of ModelProto /
// GraphProto / NodeProto / TensorProto / ValueInfoProto to load inference
// graphs without the onnx or protobuf libraries.
// Replaces ort's Session::commit_from_file parsing step
// (reference src/classifier.rs:348-350).
#pragma once
#include <cstdint>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>

namespace bn {

struct OnnxTensor {
    std::string name;
    std::vector<int64_t> dims;
    int32_t data_type = 0;  // 1=f32 6=i32 7=i64 9=bool 11=f64
    std::vector<float> f;    // decoded as float when data_type is floating
    std::vector<int64_t> i;  // decoded as int64 when data_type is integral/bool
    bool is_float() const { return data_type == 1 || data_type == 11 || data_type == 10; }
    int64_t numel() const {
        int64_t n = 1;
        for (auto d : dims) n *= d;
        return n;
    }
};

struct OnnxAttr {
    std::string name;
    int32_t type = 0;  // 1=FLOAT 2=INT 3=STRING 4=TENSOR 6=FLOATS 7=INTS
    float f = 0.f;
    int64_t i = 0;
    std::string s;
    OnnxTensor t;
    std::vector<float> floats;
    std::vector<int64_t> ints;
};

struct OnnxNode {
    std::string name, op_type, domain;
    std::vector<std::string> inputs, outputs;
    std::map<std::string, OnnxAttr> attrs;
    int64_t attr_i(const std::string &k, int64_t dflt) const {
        auto it = attrs.find(k);
        return it == attrs.end() ? dflt : it->second.i;
    }
    float attr_f(const std::string &k, float dflt) const {
        auto it = attrs.find(k);
        return it == attrs.end() ? dflt : it->second.f;
    }
    std::vector<int64_t> attr_ints(const std::string &k) const {
        auto it = attrs.find(k);
        return it == attrs.end() ? std::vector<int64_t>{} : it->second.ints;
    }
    std::string attr_s(const std::string &k, const std::string &dflt) const {
        auto it = attrs.find(k);
        return it == attrs.end() ? dflt : it->second.s;
    }
    bool has(const std::string &k) const { return attrs.count(k) != 0; }
};

struct OnnxValueInfo {
    std::string name;
    int32_t elem_type = 0;
    bool has_shape = false;
    std::vector<int64_t> shape;  // -1 for symbolic / unknown dims
};

struct OnnxModel {
    int64_t ir_version = 0;
    int64_t opset = 0;  // default-domain opset
    std::string producer;
    std::vector<OnnxNode> nodes;
    std::vector<OnnxTensor> initializers;
    std::vector<OnnxValueInfo> inputs, outputs;  // graph inputs exclude initializers
};

struct OnnxParseError : std::runtime_error {
    using std::runtime_error::runtime_error;
};

OnnxModel parse_onnx(const uint8_t *data, size_t len);
OnnxModel parse_onnx_file(const std::string &path);

}  // namespace bn

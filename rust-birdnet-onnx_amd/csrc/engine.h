// Inference engine: ONNX graph -> launch plan of hand-written HIP kernels.
// Stands in for ONNX Runtime's session (reference src/classifier.rs:340-350,
// 637-639, 721-723, 851-853): the whole numeric hot path of the reference
// (front end, CNN, head) is what this plan executes.
#pragma once
#include <hip/hip_runtime_api.h>

#include <cstdint>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "kernels.h"
#include "onnx_proto.h"

namespace bn {

struct UnsupportedModel : std::runtime_error {
    using std::runtime_error::runtime_error;
};

// Where an operand lives.
enum class Space : int32_t { NONE = 0, INPUT, ARENA, CONSTS };
struct Ref {
    Space space = Space::NONE;
    int32_t id = -1;     // ARENA: storage id; CONSTS: constant-buffer id
    int64_t offset = 0;  // elements (per sample for INPUT/ARENA)
};

enum class OpKind : int32_t { ELT, REDUCE, GEMM, CONV, DWCONV, GAP, SEFC, MBCONV, POOL, FFT };

struct PlanOp {
    OpKind kind;
    std::string name;
    Ref out, a, b, w, bias, res, scale, w2, bias2;
    Ref eb[ELT_MAX_STAGES];  // ELT: second operand of each chain stage; FFT: operands of the absorbed prologue chain
    Ref x[4];                // FFT: mel filter bank in CSR form (row starts, columns, values) and its bias
    EltDesc elt{};
    ReduceDesc red{};
    GemmDesc gemm{};
    FramePre pre{};    // GEMM with fold (LDS-resident framing kernels): per-sample chain applied while the span is loaded; operands in eb
    GemmDesc gemm2{};  // GEMM with fold (planner rule J): the product fused behind it (N == 0: none); its weights in w2, its bias in bias2
    ConvDesc conv{};
    DwDesc dw{};
    GapDesc gap{};
    SeFcDesc se{};
    MbDesc mb{};
    PoolDesc pool{};
    FftDesc fft{};
    // GEMM (gemm.se_inline, planner rule I): the squeeze-excite that PRECEDES it is computed in the GEMM's prologue: `se`
    // holds its shape, b the squeeze partial sums, x[0..3] its weights (w1, b1, w2 transposed, b2).
    // DWCONV / MBCONV: the squeeze-excite that follows is finished by the launch's last block per sample (kernels.h,
    // SeTail): `se` holds its shape, x[0..3] its weights (w1, b1, w2 transposed, b2), res the gate, scale the
    // per-sample ticket counters (a pinned arena storage of one word per sample, zero between launches)
    int32_t se_fused = 0;
    double flops_fft = 0;   // FFT: algorithmic flops per sample counted as an FFT (2.5 L log2 L per real frame) + sparse mel
    double macs = 0;        // per sample
    double macs_mfma_extra = 0;  // MBCONV: the expand part runs on the matrix cores
    double macs_recompute = 0;   // ... of which: beyond one expansion per input pixel (halo rows / columns, shared band rows, padded k)
    double macs_valu_extra = 0;  // GEMM with the squeeze-excite products in its prologue: those run on the vector ALU
    double bytes = 0;       // algorithmic bytes read+written per sample (weights excluded)
    double weight_bytes = 0;
    bool mfma = false;
};

struct Storage {
    int64_t elems = 0;  // per sample
    int32_t first = -1, last = -1;  // op indices
    bool pinned = false;            // graph output: never recycled
    bool persistent = false;        // owns its arena region for the WHOLE plan (state that must survive from launch to launch)
    int64_t arena_off = -1;         // elements per sample
};

struct OutputInfo {
    std::string name;
    std::vector<int64_t> dims;  // per sample (without batch)
    int64_t row_elems = 0;
    Ref ref;
    bool computed = false;
};

struct IoMeta {
    std::string input_name;
    std::vector<int64_t> input_shape;  // ONNX shape, -1 = dynamic
    std::vector<std::string> output_names;
    std::vector<std::vector<int64_t>> output_shapes;
};

// Immutable after build(): shared by every context of a model.
struct Plan {
    int64_t sample_count = 0;
    std::vector<PlanOp> ops;
    std::vector<Storage> storages;
    std::vector<std::vector<float>> consts;  // host copies, uploaded once per model
    std::vector<int64_t> const_off;          // element offsets inside the weights arena
    int64_t consts_elems = 0;
    int64_t arena_elems = 0;  // per sample
    std::vector<OutputInfo> outputs;
    IoMeta io;
    double macs_mfma = 0, macs_valu = 0, act_bytes = 0, weight_bytes = 0;
    // Front end counted two ways: the multiply-adds a DFT-as-matrix-product evaluation of the planned
    // filter banks performs (dft_gemm_macs: what the GEMM fallback runs) and the flops of the FFT formulation
    double dft_gemm_macs = 0, fft_flops = 0;
    // dft_performed_macs: multiply-adds the plan spends on those banks as it runs them (folded matrix product: half the
    // taps; FFT: fft flops / 2); dft_fft_equiv_flops: 2.5 L log2 L per frame of the same banks, whichever way they run
    double dft_performed_macs = 0, dft_fft_equiv_flops = 0;
    double recompute_macs = 0;  // part of macs_mfma: halo / band / padded-k recompute of the fused MBConv launches
};

// wanted_outputs: graph output indices that must be computed (others are dead code).
std::unique_ptr<Plan> build_plan(const OnnxModel &m, const std::vector<int> &wanted_outputs);
// Whether the lowering has a rule for this operator type at all (first-contact survey of a model file: a node of a mapped type can still
// be refused for its attributes or shapes, which only planning finds out; an unmapped type is refused whatever its attributes).
bool op_type_mapped(const std::string &op_type);
// Graph I/O metadata without planning (for detection before the wanted set is known).
IoMeta read_io_meta(const OnnxModel &m);

}  // namespace bn

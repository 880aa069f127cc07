# Builds libbirdnet_hip.so (gfx950 only) and the CPU oracle.
HIPCC ?= /opt/rocm/bin/hipcc
CXX := g++
ARCH ?= gfx950
PKG := rust-birdnet-onnx_amd
SRC := $(PKG)/csrc
OUT := $(PKG)/libbirdnet_hip.so
CXXFLAGS := -O3 -std=c++17 -fPIC -Wall -Wextra -Wno-unused-parameter -Iinclude
# -fno-slp-vectorize: left to itself the compiler packs neighbouring scalar f32 operations into v_pk_fma_f32 / v_pk_mul_f32 and pays
# for the pairs with register moves (mbmap's depthwise phase: 120 v_pk_fma + 216 v_mov for 240 multiply-adds); on this part a packed
# f32 instruction costs 1.56x a plain one (tools/mfma_valu_probe.cpp) and every vector instruction is paid out of the MFMA time
HIPFLAGS := $(CXXFLAGS) --offload-arch=$(ARCH) -ffp-contract=off -fno-slp-vectorize
OBJS := $(SRC)/onnx_proto.o $(SRC)/engine.o $(SRC)/detect.o $(SRC)/capi.o $(SRC)/group.o $(SRC)/host_classifier.o $(SRC)/host_capi.o $(SRC)/host_rangefilter.o $(SRC)/kernels.o $(SRC)/topk.o $(SRC)/stft.o $(SRC)/mbrow.o $(SRC)/gemm_dma.o $(SRC)/gemm_dma3.o $(SRC)/gemm_b3.o $(SRC)/mbmap.o $(SRC)/mbmap_ws.o

all: $(OUT) oracle

$(SRC)/%.o: $(SRC)/%.cpp $(wildcard $(SRC)/*.h) include/birdnet_hip.h include/birdnet_host.h
	$(CXX) $(CXXFLAGS) -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -c $< -o $@
$(SRC)/%.o: $(SRC)/%.hip $(wildcard $(SRC)/*.h)
	$(HIPCC) $(HIPFLAGS) -c $< -o $@
$(OUT): $(OBJS)
	$(HIPCC) -shared -fPIC --offload-arch=$(ARCH) -o $@ $(OBJS) -lpthread -ldl

oracle:
	$(MAKE) -C oracle

# diagnostic build: the LDS-DMA GEMM with shader-clock stamps around every K step of one block (tools/gemm_stamps.cpp; never linked into the library)
stamps: $(OUT)
	$(HIPCC) $(HIPFLAGS) -DBN_GD_STAMPS -c $(SRC)/gemm_dma.hip -o /tmp/bn_gd_stamps.o
	$(HIPCC) -O3 -std=c++17 --offload-arch=$(ARCH) -Iinclude -I$(SRC) -c tools/gemm_stamps.cpp -o /tmp/bn_gemm_stamps_main.o
	$(HIPCC) --offload-arch=$(ARCH) /tmp/bn_gemm_stamps_main.o /tmp/bn_gd_stamps.o $(SRC)/kernels.o $(SRC)/stft.o $(SRC)/topk.o $(SRC)/mbrow.o $(SRC)/mbmap.o $(SRC)/mbmap_ws.o -o tools/gemm_stamps -lpthread -ldl

# diagnostic: exact-f32 LDS-DMA GEMM against the bf16x3 form, timing + error against a double-precision product (never linked into the library)
tools/gemm3_bench: tools/gemm3_bench.cpp $(OUT)
	$(HIPCC) -O3 -std=c++17 --offload-arch=$(ARCH) -Iinclude -I$(SRC) -c tools/gemm3_bench.cpp -o /tmp/bn_gemm3_bench_main.o
	$(HIPCC) --offload-arch=$(ARCH) /tmp/bn_gemm3_bench_main.o $(SRC)/kernels.o $(SRC)/stft.o $(SRC)/topk.o $(SRC)/mbrow.o $(SRC)/mbmap.o $(SRC)/mbmap_ws.o $(SRC)/gemm_dma.o $(SRC)/gemm_dma3.o $(SRC)/gemm_b3.o -o tools/gemm3_bench -lpthread -ldl

clean:
	rm -f $(OBJS) $(OUT)
	$(MAKE) -C oracle clean
.PHONY: all oracle clean stamps

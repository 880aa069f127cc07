# Builds libbirdnet_hip.so (gfx950 only) and the CPU oracle.
HIPCC ?= /opt/rocm/bin/hipcc
CXX := g++
ARCH ?= gfx950
PKG := rust-birdnet-onnx_amd
SRC := $(PKG)/csrc
OUT := $(PKG)/libbirdnet_hip.so
CXXFLAGS := -O3 -std=c++17 -fPIC -Wall -Wextra -Wno-unused-parameter -Iinclude
HIPFLAGS := $(CXXFLAGS) --offload-arch=$(ARCH) -ffp-contract=off
OBJS := $(SRC)/onnx_proto.o $(SRC)/engine.o $(SRC)/detect.o $(SRC)/capi.o $(SRC)/group.o $(SRC)/host_classifier.o $(SRC)/host_capi.o $(SRC)/host_rangefilter.o $(SRC)/kernels.o $(SRC)/topk.o $(SRC)/stft.o $(SRC)/mbrow.o $(SRC)/gemm_dma.o $(SRC)/mbmap.o

all: $(OUT) oracle

$(SRC)/%.o: $(SRC)/%.cpp $(wildcard $(SRC)/*.h) include/birdnet_hip.h include/birdnet_host.h
	$(CXX) $(CXXFLAGS) -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -c $< -o $@
$(SRC)/%.o: $(SRC)/%.hip $(wildcard $(SRC)/*.h)
	$(HIPCC) $(HIPFLAGS) -c $< -o $@
$(OUT): $(OBJS)
	$(HIPCC) -shared -fPIC --offload-arch=$(ARCH) -o $@ $(OBJS) -lpthread -ldl

oracle:
	$(MAKE) -C oracle

clean:
	rm -f $(OBJS) $(OUT)
	$(MAKE) -C oracle clean
.PHONY: all oracle clean

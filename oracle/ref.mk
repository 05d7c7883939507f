# oracle/ref.mk — TEST INFRASTRUCTURE.  Compiles the REFERENCE's own test/driver sources from where
# they lie under $(REF) (never copied), unmodified, against OUR library + the stand-in generated headers
# in include/aot/.  Outputs go to oracle/_ref/ only (git-ignored; they travel to the GPU box).
# These binaries are the drop-in proof: a driver written for the reference's AOT objects runs on libhlmi.so.
CXX      ?= g++
RT       := $(REF)/src/runtime
TOOLS    := $(REF)/tools
OUTDIR   := $(ROOT)/oracle/_ref
LIBDIR   := $(ROOT)/halide_amd/lib
CXXFLAGS := -std=c++17 -O2 -fopenmp -DHALIDE_NO_PNG -DHALIDE_NO_JPEG -I$(RT) -I$(TOOLS) -I$(ROOT)/include/aot
LDFLAGS  := -L$(LIBDIR) -lhlmi -Wl,-rpath,'$$ORIGIN/../../halide_amd/lib' -lpthread -ldl

TARGETS := $(OUTDIR)/blur_test $(OUTDIR)/local_laplacian_process $(OUTDIR)/bilateral_grid_filter \
           $(OUTDIR)/nl_means_process $(OUTDIR)/stencil_chain_process $(OUTDIR)/conv_layer_process $(OUTDIR)/camera_pipe_process \
           $(OUTDIR)/depthwise_separable_conv_process $(OUTDIR)/unsharp_filter $(OUTDIR)/max_filter_filter $(OUTDIR)/hist_filter $(OUTDIR)/harris_filter $(OUTDIR)/iir_blur_filter $(OUTDIR)/interpolate_filter $(OUTDIR)/lens_blur_process $(OUTDIR)/bgu_filter

# The reference's own RunGen (tools/RunGenMain.cpp + RunGen.h, compiled unmodified) linked with the per-pipeline
# registration unit of tests/cpp/rungen_registration.cpp: the reference's consumer of <name>_argv / <name>_metadata /
# the bounds-query protocol, one <name>.rungen per pipeline as in the reference's build (apps/*/Makefile, *.rungen).
RUNGEN_PIPELINES := local_laplacian bilateral_grid halide_blur nl_means stencil_chain conv_layer camera_pipe \
                    depthwise_separable_conv unsharp max_filter hist harris interpolate iir_blur lens_blur bgu
TARGETS += $(patsubst %,$(OUTDIR)/%.rungen,$(RUNGEN_PIPELINES))
# own test programs that need the reference's headers (tests/cpp/*.cpp; sources are ours, headers the reference's)
TARGETS += $(OUTDIR)/entry_protocol_ref $(OUTDIR)/device_interface_test
# the PNG path of tools/halide_image_io.h: two drivers built WITHOUT -DHALIDE_NO_PNG against tests/cpp/png_shim/png.h (the libpng
# calls the reference makes, over zlib)
PNGFLAGS := $(filter-out -DHALIDE_NO_PNG,$(CXXFLAGS)) -I$(ROOT)/tests/cpp/png_shim
TARGETS += $(OUTDIR)/interpolate_filter_png $(OUTDIR)/local_laplacian_process_png

all: $(TARGETS)

$(OUTDIR)/RunGenMain.o: $(TOOLS)/RunGenMain.cpp $(TOOLS)/RunGen.h
	$(CXX) $(CXXFLAGS) -c $< -o $@
$(OUTDIR)/%.rungen: $(OUTDIR)/RunGenMain.o $(ROOT)/tests/cpp/rungen_registration.cpp $(LIBDIR)/libhlmi.so
	$(CXX) $(CXXFLAGS) -DPIPELINE=$* $(OUTDIR)/RunGenMain.o $(ROOT)/tests/cpp/rungen_registration.cpp -o $@ $(LDFLAGS)
$(OUTDIR)/entry_protocol_ref: $(ROOT)/tests/cpp/entry_protocol_ref.cpp $(LIBDIR)/libhlmi.so
	$(CXX) $(CXXFLAGS) $< -o $@ $(LDFLAGS)
$(OUTDIR)/device_interface_test: $(ROOT)/tests/cpp/device_interface_test.cpp $(LIBDIR)/libhlmi.so
	$(CXX) $(CXXFLAGS) $< -o $@ $(LDFLAGS)

# apps/blur/test.cpp: compares halide_blur() with its own scalar + SSE2 loops, prints "Success!"
$(OUTDIR)/blur_test: $(REF)/apps/blur/test.cpp $(LIBDIR)/libhlmi.so
	$(CXX) $(CXXFLAGS) -msse2 $< -o $@ $(LDFLAGS)

# apps/local_laplacian/process.cpp: load image -> local_laplacian(+_auto_schedule) -> benchmark -> save
$(OUTDIR)/local_laplacian_process: $(REF)/apps/local_laplacian/process.cpp $(LIBDIR)/libhlmi.so
	$(CXX) $(CXXFLAGS) $< -o $@ $(LDFLAGS)

# the other app drivers, same recipe (apps/<app>/{filter,process}.cpp, unmodified)
$(OUTDIR)/bilateral_grid_filter: $(REF)/apps/bilateral_grid/filter.cpp $(LIBDIR)/libhlmi.so
	$(CXX) $(CXXFLAGS) $< -o $@ $(LDFLAGS)
$(OUTDIR)/nl_means_process: $(REF)/apps/nl_means/process.cpp $(LIBDIR)/libhlmi.so
	$(CXX) $(CXXFLAGS) $< -o $@ $(LDFLAGS)
$(OUTDIR)/stencil_chain_process: $(REF)/apps/stencil_chain/process.cpp $(LIBDIR)/libhlmi.so
	$(CXX) $(CXXFLAGS) $< -o $@ $(LDFLAGS)
$(OUTDIR)/conv_layer_process: $(REF)/apps/conv_layer/process.cpp $(LIBDIR)/libhlmi.so
	$(CXX) $(CXXFLAGS) $< -o $@ $(LDFLAGS)
$(OUTDIR)/camera_pipe_process: $(REF)/apps/camera_pipe/process.cpp $(LIBDIR)/libhlmi.so
	$(CXX) $(CXXFLAGS) $< -o $@ $(LDFLAGS)
$(OUTDIR)/depthwise_separable_conv_process: $(REF)/apps/depthwise_separable_conv/process.cpp $(LIBDIR)/libhlmi.so
	$(CXX) $(CXXFLAGS) $< -o $@ $(LDFLAGS)
$(OUTDIR)/unsharp_filter: $(REF)/apps/unsharp/filter.cpp $(LIBDIR)/libhlmi.so
	$(CXX) $(CXXFLAGS) $< -o $@ $(LDFLAGS)
$(OUTDIR)/max_filter_filter: $(REF)/apps/max_filter/filter.cpp $(LIBDIR)/libhlmi.so
	$(CXX) $(CXXFLAGS) $< -o $@ $(LDFLAGS)
$(OUTDIR)/hist_filter: $(REF)/apps/hist/filter.cpp $(LIBDIR)/libhlmi.so
	$(CXX) $(CXXFLAGS) $< -o $@ $(LDFLAGS)
$(OUTDIR)/harris_filter: $(REF)/apps/harris/filter.cpp $(LIBDIR)/libhlmi.so
	$(CXX) $(CXXFLAGS) $< -o $@ $(LDFLAGS)
$(OUTDIR)/iir_blur_filter: $(REF)/apps/iir_blur/filter.cpp $(LIBDIR)/libhlmi.so
	$(CXX) $(CXXFLAGS) $< -o $@ $(LDFLAGS)
$(OUTDIR)/lens_blur_process: $(REF)/apps/lens_blur/process.cpp $(LIBDIR)/libhlmi.so
	$(CXX) $(CXXFLAGS) $< -o $@ $(LDFLAGS)
$(OUTDIR)/bgu_filter: $(REF)/apps/bgu/filter.cpp $(LIBDIR)/libhlmi.so
	$(CXX) $(CXXFLAGS) $< -o $@ $(LDFLAGS)
$(OUTDIR)/interpolate_filter_png: $(REF)/apps/interpolate/filter.cpp $(ROOT)/tests/cpp/png_shim/png.h $(LIBDIR)/libhlmi.so
	$(CXX) $(PNGFLAGS) $< -o $@ $(LDFLAGS) -lz
$(OUTDIR)/local_laplacian_process_png: $(REF)/apps/local_laplacian/process.cpp $(ROOT)/tests/cpp/png_shim/png.h $(LIBDIR)/libhlmi.so
	$(CXX) $(PNGFLAGS) $< -o $@ $(LDFLAGS) -lz
# RGBA input: without libpng the driver is fed the reference's own 4-dimensional ".tmp" format (tools/halide_image_io.h:1630-1720)
$(OUTDIR)/interpolate_filter: $(REF)/apps/interpolate/filter.cpp $(LIBDIR)/libhlmi.so
	$(CXX) $(CXXFLAGS) $< -o $@ $(LDFLAGS)

// Stand-in for TensorRT's <NvInfer.h> (test infrastructure): the names d2frontend/CNN/superpoint_tensorrt.h declares members with.
#pragma once
#include <cstdint>
#include <memory>
namespace nvinfer1 {
struct Dims { int32_t nbDims = 0; int32_t d[8] = {0, 0, 0, 0, 0, 0, 0, 0}; };
class IBuilder; class INetworkDefinition; class IBuilderConfig; class ICudaEngine; class IExecutionContext;
}
namespace tensorrt_common { template <class T> using TensorRTUniquePtr = std::unique_ptr<T>; }

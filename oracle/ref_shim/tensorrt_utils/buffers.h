// Stand-in for tensorrt_utils/buffers.h (test infrastructure): host-side output buffers by tensor name.
#pragma once
#include <map>
#include <string>
namespace tensorrt_buffer {
class BufferManager {
 public:
  std::map<std::string, void*> host;
  void* getHostBuffer(const std::string& name) const { auto it = host.find(name); return it == host.end() ? nullptr : it->second; }
};
}

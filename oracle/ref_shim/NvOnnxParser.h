#pragma once
namespace nvonnxparser { class IParser; }

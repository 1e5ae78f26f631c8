#pragma once
namespace D2Common { namespace Utility { struct TicToc { double toc() { return 0.0; } }; } }

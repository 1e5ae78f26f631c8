#pragma once
namespace spdlog { template <class... A> inline void debug(A&&...) {} template <class... A> inline void error(A&&...) {} template <class... A> inline void info(A&&...) {} }
#define SPDLOG_DEBUG(...) ((void)0)
#define SPDLOG_INFO(...) ((void)0)
#define SPDLOG_ERROR(...) ((void)0)

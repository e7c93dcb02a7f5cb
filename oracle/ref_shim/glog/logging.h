// TEST INFRASTRUCTURE: glog stand-in for compiling the reference's headers (messages are discarded).
#pragma once
#include <iostream>
#include <sstream>
struct ShimNullLog {
  template <typename T> ShimNullLog& operator<<(const T&) { return *this; }
  ShimNullLog& operator<<(std::ostream& (*)(std::ostream&)) { return *this; }
};
#define LOG(x) ShimNullLog()
#define VLOG(x) ShimNullLog()
#define LOG_IF(x, c) ShimNullLog()
#define CHECK(x) ShimNullLog()
#define CHECK_EQ(a, b) ShimNullLog()
#define CHECK_NE(a, b) ShimNullLog()

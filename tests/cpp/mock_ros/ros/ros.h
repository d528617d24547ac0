// mock roscpp: declarations only (see ../README.md)
#pragma once
#include <boost_shared_ptr_mock.h>
#include <cstdint>
#include <iostream>
#include <sstream>
#include <string>
namespace ros {
struct Time { uint32_t sec = 0, nsec = 0; Time() {} explicit Time(double) {} static Time now() { return Time(); } double toSec() const { return sec + 1e-9 * nsec; } };
struct Duration { explicit Duration(double = 0.0) {} };
struct TimerEvent {};
struct Timer {};
struct TransportHints { TransportHints& tcpNoDelay(bool = true) { return *this; } };
struct Subscriber {};
struct Publisher {
  template <class M> void publish(const M&) const {}
  uint32_t getNumSubscribers() const { return 0; }
};
struct NodeHandle {
  template <class T> bool param(const std::string&, T& v, const T& d) const { v = d; return false; }
  template <class T, class D> bool param(const std::string&, T& v, const D& d) const { v = T(d); return false; }
  template <class M, class C> Subscriber subscribe(const std::string&, uint32_t, void (C::*)(const boost::shared_ptr<M const>&), C*, const TransportHints& = TransportHints()) { return Subscriber(); }
  template <class M> Publisher advertise(const std::string&, uint32_t, bool = false) { return Publisher(); }
  template <class C> Timer createTimer(Duration, void (C::*)(const TimerEvent&), C*, bool = false, bool = true) { return Timer(); }
};
inline void init(int&, char**, const std::string&) {}
inline void shutdown() {}
inline void spin() {}
struct MultiThreadedSpinner { explicit MultiThreadedSpinner(uint32_t = 0) {} void spin() {} };
}  // namespace ros
#define ROS_ERROR_STREAM(x) do { std::ostringstream ros_mock_ss; ros_mock_ss << x; } while (0)
#define ROS_ERROR(...) do { } while (0)
#define ROS_INFO(...) do { } while (0)

#pragma once
#include <vector>
namespace std_msgs { struct Float64MultiArray { std::vector<double> data; }; }

#pragma once
#include <tf/transform_broadcaster.h>

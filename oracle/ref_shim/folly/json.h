#pragma once
#include "dynamic.h"

// TEST INFRASTRUCTURE: PixFlow.h:15 includes calibration/KeypointMatchers.h without using it (a dead include).
#pragma once

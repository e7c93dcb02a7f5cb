// TEST INFRASTRUCTURE: forced include for the reference's program sources (-include): names they expect from headers
// that are not compiled here.
#pragma once
namespace surround360 { namespace calibration {} }

// What RayTracer_mi355x.cpp offers a host beyond the reference's own class declaration.
#pragma once

#include "../../include/rt_mi355x.h"

class RayTracer; // the reference's class, src/rt/RayTracer.h:34-94

// The library handle behind a created RayTracer (NULL before create and after destroy).  Through it a host reaches
// what the reference's class has no method for: the image to present (rt_output_device_ptr, rt_read_output), batched
// and sharded renders, other scenes, the HIP stream (rt_mi355x_set_stream below).
rt_handle rt_mi355x_handle(const RayTracer* tracer);

// GPU that RayTracer::create uses from now on (default 0) and the HIP stream that RayTracer::render of this object
// enqueues on (default: the null stream).
void rt_mi355x_set_device(int device_id);
void rt_mi355x_set_stream(const RayTracer* tracer, void* hip_stream);

/* Diagnostics convention of the reference (reference util.h:4): one line on
 * stderr tagged with file, line and function.  Errors are reported this way
 * and turned into a false/NULL return; the library never aborts. */
#pragma once
#include <stdio.h>
#define MSG(fmt, ...) \
    fprintf(stderr, "%s(%d) at %s(): " fmt "\n", __FILE__, __LINE__, __func__, ##__VA_ARGS__)

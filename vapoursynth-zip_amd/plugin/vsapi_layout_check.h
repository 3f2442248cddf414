/* Compile-time pin of the VSAPI function table (VapourSynth API 4): every member libvszip.so or the test host calls sits at
 * the slot the public VapourSynth4.h gives it. The indices are written ONCE here, from the public header's member order
 * (createVideoFilter = 0 ... removeLogHandler = 105 in API 4.0; later minor versions only append), so a future edit of
 * plugin/VapourSynth4_min.h — our own declaration, VapourSynth is not in the build image — cannot silently shift the table:
 * a plugin built against a shifted table would call the wrong host function. Included by vszip_plugin.cpp and
 * tests/fakevs/fakevs.cpp after the API header; it also checks the REAL header when built with -DVSZIP_USE_SYSTEM_VS_HEADER.
 * (SURVEY section 7 hard part 6: no real core exists here to run under.) */
#ifndef VSZIP_VSAPI_LAYOUT_CHECK_H
#define VSZIP_VSAPI_LAYOUT_CHECK_H
#include <cstddef>

#define VSZIP_VSAPI_SLOT(member, index) \
    static_assert(offsetof(VSAPI, member) == (index) * sizeof(void *), "VSAPI::" #member " is not at slot " #index " of the API-4 table")

VSZIP_VSAPI_SLOT(createVideoFilter, 0);
VSZIP_VSAPI_SLOT(createVideoFilter2, 1);
VSZIP_VSAPI_SLOT(freeNode, 7);
VSZIP_VSAPI_SLOT(addNodeRef, 8);
VSZIP_VSAPI_SLOT(getNodeType, 9);
VSZIP_VSAPI_SLOT(getVideoInfo, 10);
VSZIP_VSAPI_SLOT(newVideoFrame, 12);
VSZIP_VSAPI_SLOT(newVideoFrame2, 13);
VSZIP_VSAPI_SLOT(freeFrame, 16);
VSZIP_VSAPI_SLOT(addFrameRef, 17);
VSZIP_VSAPI_SLOT(copyFrame, 18);
VSZIP_VSAPI_SLOT(getFramePropertiesRO, 19);
VSZIP_VSAPI_SLOT(getFramePropertiesRW, 20);
VSZIP_VSAPI_SLOT(getStride, 21);
VSZIP_VSAPI_SLOT(getReadPtr, 22);
VSZIP_VSAPI_SLOT(getWritePtr, 23);
VSZIP_VSAPI_SLOT(getVideoFrameFormat, 24);
VSZIP_VSAPI_SLOT(getFrameType, 26);
VSZIP_VSAPI_SLOT(getFrameWidth, 27);
VSZIP_VSAPI_SLOT(getFrameHeight, 28);
VSZIP_VSAPI_SLOT(queryVideoFormat, 32);
VSZIP_VSAPI_SLOT(queryVideoFormatID, 34);
VSZIP_VSAPI_SLOT(getVideoFormatByID, 35);
VSZIP_VSAPI_SLOT(getFrame, 36);
VSZIP_VSAPI_SLOT(getFrameFilter, 38);
VSZIP_VSAPI_SLOT(requestFrameFilter, 39);
VSZIP_VSAPI_SLOT(setFilterError, 42);
VSZIP_VSAPI_SLOT(createMap, 47);
VSZIP_VSAPI_SLOT(freeMap, 48);
VSZIP_VSAPI_SLOT(clearMap, 49);
VSZIP_VSAPI_SLOT(copyMap, 50);
VSZIP_VSAPI_SLOT(mapSetError, 51);
VSZIP_VSAPI_SLOT(mapGetError, 52);
VSZIP_VSAPI_SLOT(mapNumKeys, 53);
VSZIP_VSAPI_SLOT(mapGetKey, 54);
VSZIP_VSAPI_SLOT(mapDeleteKey, 55);
VSZIP_VSAPI_SLOT(mapNumElements, 56);
VSZIP_VSAPI_SLOT(mapGetType, 57);
VSZIP_VSAPI_SLOT(mapGetInt, 59);
VSZIP_VSAPI_SLOT(mapGetIntArray, 61);
VSZIP_VSAPI_SLOT(mapSetInt, 62);
VSZIP_VSAPI_SLOT(mapGetFloat, 64);
VSZIP_VSAPI_SLOT(mapGetFloatArray, 66);
VSZIP_VSAPI_SLOT(mapSetFloat, 67);
VSZIP_VSAPI_SLOT(mapGetData, 69);
VSZIP_VSAPI_SLOT(mapGetDataSize, 70);
VSZIP_VSAPI_SLOT(mapSetData, 72);
VSZIP_VSAPI_SLOT(mapGetNode, 73);
VSZIP_VSAPI_SLOT(mapSetNode, 74);
VSZIP_VSAPI_SLOT(mapConsumeNode, 75);
VSZIP_VSAPI_SLOT(registerFunction, 82);
VSZIP_VSAPI_SLOT(getPluginByID, 83);
VSZIP_VSAPI_SLOT(getPluginByNamespace, 84);
VSZIP_VSAPI_SLOT(invoke, 96);
VSZIP_VSAPI_SLOT(getAPIVersion, 102);
VSZIP_VSAPI_SLOT(logMessage, 103);
/* VSPLUGINAPI (what VapourSynthPluginInit2 receives): getAPIVersion, configPlugin, registerFunction */
static_assert(offsetof(VSPLUGINAPI, getAPIVersion) == 0 && offsetof(VSPLUGINAPI, configPlugin) == sizeof(void *) && offsetof(VSPLUGINAPI, registerFunction) == 2 * sizeof(void *),
              "VSPLUGINAPI layout");
#undef VSZIP_VSAPI_SLOT
#endif

// gr_stub: see block.h
#pragma once
#define __GR_ATTR_EXPORT
#define __GR_ATTR_IMPORT

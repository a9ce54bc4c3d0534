/* stand-in for <epoxy/gl.h>: the names reference standalone.c:65-85 uses (its GLUT window mode) */
#pragma once
typedef unsigned int GLenum;
#define GL_CW             0x0900
#define GL_CCW            0x0901
#define GL_FRONT_AND_BACK 0x0408
#define GL_POINT          0x1B00
#define GL_LINE           0x1B01
#define GL_FILL           0x1B02
void glPolygonMode(GLenum face, GLenum mode);
void glFrontFace(GLenum mode);
